#!/bin/bash
# round 4, session L: vote verification on slot bitmaps -- vote / round tests, full-size C3, one whole round under the knobs, bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "vote or round_decides or no_quorum or classic or c4_shaped or full_size_c3 or two_ranks" > gpurun_out/pytest_gpu_l.log 2>&1; tail -3 gpurun_out/pytest_gpu_l.log
timeout 400 python scripts/step_ab.py C3b 50 > gpurun_out/step_ab.log 2>&1; tail -18 gpurun_out/step_ab.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_l.json 2> gpurun_out/bench_l.err; tail -c 1500 gpurun_out/bench_l.json
