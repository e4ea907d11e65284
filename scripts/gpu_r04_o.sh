#!/bin/bash
# round 4, session O: view change without intermediate waits (pinned staging, configuration id polled from the mapped page) -- view /
# churn / streaming parity, apply_cut timed at 10^4 and 10^6 members, bench with time-to-stable-cut as a median of five trials
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "view or churn or round or streaming or q4 or joiners or bulk or config_id or timers or failure_detector" > gpurun_out/pytest_gpu_o.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" gpurun_out/pytest_gpu_o.log | tail -4
timeout 100 python scripts/time_apply.py 10000 > gpurun_out/time_apply_10k.txt 2>&1; grep round gpurun_out/time_apply_10k.txt
timeout 300 python scripts/time_apply.py 1000000 > gpurun_out/time_apply_1m.txt 2>&1; grep round gpurun_out/time_apply_1m.txt
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras > gpurun_out/bench_o.json 2> gpurun_out/bench_o.err; tail -c 300 gpurun_out/bench_o.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_o.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_min", "time_to_stable_cut_ms", "time_to_stable_cut_trials_ms", "decided", "cut_size")})
PY
