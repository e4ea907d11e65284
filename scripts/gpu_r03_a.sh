#!/bin/bash
# Round-3 GPU session A: parity tests, A/B of the window-set count, waves sweep, bench.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
timeout 300 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -12 gpurun_out/ab.log
timeout 300 python scripts/waves_sweep.py C3b 20 8,10,12,14,15,16 > gpurun_out/waves.log 2>&1; tail -8 gpurun_out/waves.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 3000 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
