#!/bin/bash
# round 4, session E: GPU suite on the product / test split + Q4 memo, bench, C5 rounds at 10^6 with the view-change stages
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 400 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_c3b.json'))
    print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','ms_per_step_median','time_to_stable_cut_ms','decided','cut_size','load_from_host_ms')})
    print({k:d['roofline'][k] for k in ('frac','kernel_ms','traffic_over_bytes','kernel_ms_filter_per_delivery')})
except Exception as e:
    print("bench line unreadable", e)
PY
timeout 900 python scripts/c5_stream.py 1000000 4 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5.err; tail -c 300 gpurun_out/c5.err
python - <<'PY'
import json
for ln in open('gpurun_out/c5_1m.jsonl'):
    d=json.loads(ln); print({k:d.get(k) for k in ('round','kernel_ms','kernel_frac_of_8TBps','round_from_boundary_ms','apply_cut_ms','waves_per_workgroup','q4_at_risk')})
PY
