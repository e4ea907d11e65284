#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python scripts/c5_stream.py 1000000 2 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5_1m.err; cat gpurun_out/c5_1m.jsonl | cut -c1-520; tail -2 gpurun_out/c5_1m.err
timeout 300 python scripts/c4_shard.py > gpurun_out/c4.log 2>&1; tail -4 gpurun_out/c4.log | cut -c1-900
