#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 400 python scripts/blocks_per_cu.py C3b 20 > gpurun_out/bpc.log 2>&1; tail -8 gpurun_out/bpc.log
timeout 300 python scripts/waves_sweep.py C3b 20 12,14,15,16 > gpurun_out/waves.log 2>&1; tail -5 gpurun_out/waves.log
