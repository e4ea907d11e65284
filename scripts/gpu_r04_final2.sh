#!/bin/bash
# round 4, closing session: the whole GPU suite, smoke, the bench as the driver runs it, rocprofv3 kernel statistics of the bench command
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | tail -6
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 300 gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c3b.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','ms_per_step_median','time_to_stable_cut_ms','time_to_stable_cut_trials_ms','decided','cut_size')})
print({k:d['roofline'][k] for k in ('frac','kernel_ms','traffic_over_bytes','kernel_ms_filter_per_delivery')}, d['round_index']['index_build_ms'], d['cpu_baseline'])
PY
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -o bench -- \
    python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-extras > "$R/gpurun_out/prof_bench.log" 2>&1
cd "$R"
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-150
python scripts/trace_gaps.py gpurun_out/prof_bench > gpurun_out/trace_gaps.txt 2>&1; tail -9 gpurun_out/trace_gaps.txt
