#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
RAPID_MI355X_LIB=$PWD/rapid_amd/librapid_mi355x_lean.so timeout 500 python scripts/blocks_per_cu.py C3b 20 > gpurun_out/bpc.log 2>&1; tail -9 gpurun_out/bpc.log
