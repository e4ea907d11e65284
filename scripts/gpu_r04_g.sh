#!/bin/bash
# round 4, session G: index build time with / without the observer memo, A/B against the previous build, the handed-over test, bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python scripts/index_probe.py C3b 2>&1 | tail -4
timeout 400 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -4 gpurun_out/ab.log
timeout 600 python -m pytest tests -m gpu -q -k "handed_over or q4 or two_processes" > gpurun_out/pytest_gpu_g.log 2>&1; tail -3 gpurun_out/pytest_gpu_g.log
timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-extras > gpurun_out/bench_c3b_g.json 2> gpurun_out/bench.err; tail -c 400 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_c3b_g.json'))
    print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','ms_per_step_median','time_to_stable_cut_ms','decided','cut_size')})
    print({k:d['roofline'][k] for k in ('frac','kernel_ms')}, d['round_index']['index_build_ms'])
except Exception as e:
    print("bench line unreadable", e)
PY
