#!/bin/bash
# one A/B session: every rapid_amd/librapid_mi355x_<tag>.so present against the default build (scripts/ab_variants.py)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 500 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -22 gpurun_out/ab.log
