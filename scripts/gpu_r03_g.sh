#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 500 python scripts/stagger_sweep.py C3b 20 > gpurun_out/stagger.log 2>&1; tail -9 gpurun_out/stagger.log
