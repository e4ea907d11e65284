#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 400 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -18 gpurun_out/ab.log
