#!/bin/bash
# round 4, session B: load-shape variants of the boundary kernel, waves sweep with the stream-only floor, SQ counters, the three fixed tests
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -12 gpurun_out/ab.log
timeout 300 python scripts/waves_sweep.py C3b 20 10,12,15 > gpurun_out/waves.log 2>&1; cat gpurun_out/waves.log
bash scripts/pmc_sq.sh > gpurun_out/sq_summary.txt 2>&1; tail -24 gpurun_out/sq_summary.txt
timeout 900 python -m pytest tests -m gpu -q -k "declared_alert_set or generated or one_million" > gpurun_out/pytest_gpu_b.log 2>&1; tail -5 gpurun_out/pytest_gpu_b.log
