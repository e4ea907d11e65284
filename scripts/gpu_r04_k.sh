#!/bin/bash
# round 4, session K: waves per workgroup (exact), pool size, one whole round under the index knob, vote tests, bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "vote or round_decides or no_quorum or classic or c4_shaped" > gpurun_out/pytest_gpu_k.log 2>&1; tail -3 gpurun_out/pytest_gpu_k.log
timeout 400 python scripts/waves_sweep.py C3b 20 13,14,15,16 > gpurun_out/waves.log 2>&1; cat gpurun_out/waves.log | tail -5
timeout 400 python scripts/pool_ab.py C3b 20 > gpurun_out/pool.log 2>&1; tail -5 gpurun_out/pool.log
timeout 400 python scripts/step_ab.py C3b 50 > gpurun_out/step_ab.log 2>&1; tail -16 gpurun_out/step_ab.log
