#!/bin/bash
# round 4, session J: the one-launch round index and the entry staging -- parity subset, index probe, bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "tables_in_memory or handed_over or generated or population_scenarios or declared or q4 or round_decides or full_size_c3 or churn or smoke or c4_shaped" > gpurun_out/pytest_gpu_j.log 2>&1; tail -5 gpurun_out/pytest_gpu_j.log
timeout 300 python scripts/index_probe.py C3b 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-pmc > gpurun_out/bench_c3b_j.json 2> gpurun_out/bench.err; tail -c 300 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_c3b_j.json'))
    print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','ms_per_step_median','time_to_stable_cut_ms','decided','cut_size')})
    print({k:d['roofline'][k] for k in ('frac','kernel_ms','kernel_ms_filter_per_delivery')}, d['round_index']['index_build_ms'])
    print(d.get('generated_streams'))
except Exception as e:
    print("bench line unreadable", e)
PY
timeout 300 python scripts/rounds_probe.py 10 > gpurun_out/rounds_probe.txt 2>&1; tail -7 gpurun_out/rounds_probe.txt
