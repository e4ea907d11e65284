#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
RAPID_MI355X_LIB=$PWD/rapid_amd/librapid_mi355x_sets6.so timeout 600 python scripts/c5_stream.py 1000000 2 1024 2>/dev/null | cut -c1-330
