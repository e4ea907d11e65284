#!/bin/bash
# round 4, session M: packed rounds with four / six window sets and at most eight waves; dictionary in memory gathered a window ahead
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "memory_mode or million or 100k or streaming or generated or c4_" --durations=8 > gpurun_out/pytest_gpu_m.log 2>&1; tail -14 gpurun_out/pytest_gpu_m.log
timeout 600 python scripts/c5_stream.py 1000000 3 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5.err; tail -c 300 gpurun_out/c5.err; cut -c1-700 gpurun_out/c5_1m.jsonl
