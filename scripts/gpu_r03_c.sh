#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -14 gpurun_out/ab.log
timeout 600 bash scripts/phase_timers.sh C3b > gpurun_out/timers.log 2>&1; grep -v "^receivers  " gpurun_out/timers.log | tail -14
