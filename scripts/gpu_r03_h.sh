#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error\|error" gpurun_out/pytest_gpu.log | tail -8
timeout 600 python scripts/c5_stream.py 1000000 3 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5_1m.err; cat gpurun_out/c5_1m.jsonl | cut -c1-900; tail -3 gpurun_out/c5_1m.err
