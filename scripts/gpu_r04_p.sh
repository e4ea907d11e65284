#!/bin/bash
# round 4, session P: view change with the member flags cleared by the memo kernel and a one-workgroup configuration id for small
# views -- parity of the view / churn tests, apply_cut at 10^4 / 10^6 (with the phase breakdown at 10^6), bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "view or churn or round_decides or streaming_rounds_with or q4 or joiners or bulk or config_id or 100k" > gpurun_out/pytest_gpu_p.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" gpurun_out/pytest_gpu_p.log | tail -3
timeout 100 python scripts/time_apply.py 10000 > gpurun_out/time_apply_10k.txt 2>&1; grep round gpurun_out/time_apply_10k.txt
timeout 300 python scripts/time_apply.py 1000000 > gpurun_out/time_apply_1m.txt 2>&1; grep round gpurun_out/time_apply_1m.txt
RAPID_TIME_VIEW=1 timeout 300 python scripts/time_apply.py 1000000 > gpurun_out/time_apply_1m_phases.txt 2>&1; tail -7 gpurun_out/time_apply_1m_phases.txt
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras > gpurun_out/bench_p.json 2> gpurun_out/bench_p.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_p.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_min", "time_to_stable_cut_ms", "time_to_stable_cut_trials_ms", "decided", "cut_size")})
PY
