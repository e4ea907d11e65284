#!/bin/bash
# round 4, session H: where the boundary kernel's time goes -- rounds probe, phase timers, SQ counters
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 400 python scripts/rounds_probe.py 10 > gpurun_out/rounds_probe.txt 2>&1; cat gpurun_out/rounds_probe.txt | tail -8
RAPID_MI355X_LIB="$PWD/rapid_amd/librapid_mi355x_timers.so" timeout 300 python scripts/phase_timers.py C3b > gpurun_out/phase_timers.txt 2>&1; head -14 gpurun_out/phase_timers.txt
bash scripts/pmc_sq.sh > gpurun_out/sq_summary.txt 2>&1; tail -24 gpurun_out/sq_summary.txt
