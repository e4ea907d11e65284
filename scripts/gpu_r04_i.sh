#!/bin/bash
# round 4, session I: the whole GPU suite on the current build, A/B of the prepared variants (lean open off, three window sets,
# 2 x 10 waves per CU at 96 VGPRs, cold windows outside the turn loop), bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
timeout 500 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -9 gpurun_out/ab.log
timeout 600 python bench.py > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 300 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_c3b.json'))
    print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','ms_per_step_median','time_to_stable_cut_ms','decided','cut_size')})
    print({k:d['roofline'][k] for k in ('frac','kernel_ms','traffic_over_bytes')}, d['round_index']['index_build_ms'])
except Exception as e:
    print("bench line unreadable", e)
PY
