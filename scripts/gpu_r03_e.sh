#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -3
timeout 600 bash scripts/phase_timers.sh C3b > gpurun_out/timers.log 2>&1; grep -v "^receivers  " gpurun_out/timers.log | tail -12
rm -f rapid_amd/librapid_mi355x_timers.so
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 3800 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
