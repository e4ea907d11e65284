#!/bin/bash
# round 4, session A: the whole GPU suite on the one-pass boundary path, smoke, a first bench line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | tail -30
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.err; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_c3b.json'))
    print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','time_to_stable_cut_ms','decided','cut_size','load_from_host_ms')})
    print(d['roofline'])
    print(d['generated_streams'])
    print(d['round_index'])
except Exception as e:
    print("bench line unreadable", e)
PY
