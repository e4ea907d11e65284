#!/bin/bash
# round 4, session C: what the memory system delivers for 20-byte records by load shape; cache-policy variants of the kernel
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 scripts/micro/boundary_shapes > gpurun_out/boundary_shapes.txt 2>&1; cat gpurun_out/boundary_shapes.txt
timeout 600 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -8 gpurun_out/ab.log
