#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 bash scripts/phase_timers.sh C3b > gpurun_out/timers.log 2>&1; cat gpurun_out/timers.log | tail -40
