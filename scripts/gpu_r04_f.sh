#!/bin/bash
# round 4, session F: suite, bench, A/B against the previous commit's build, C5 rounds at 10^6 (packed detector), kernel trace of the bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python scripts/ab_variants.py C3b 3 20 > gpurun_out/ab.log 2>&1; tail -4 gpurun_out/ab.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 400 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_c3b.json'))
    print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','ms_per_step_median','time_to_stable_cut_ms','decided','cut_size')})
    print({k:d['roofline'][k] for k in ('frac','kernel_ms','traffic_over_bytes','kernel_ms_filter_per_delivery')}, d['round_index'])
except Exception as e:
    print("bench line unreadable", e)
PY
timeout 900 python scripts/c5_stream.py 1000000 3 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5.err; tail -c 300 gpurun_out/c5.err
python - <<'PY'
import json
for ln in open('gpurun_out/c5_1m.jsonl'):
    d=json.loads(ln); print({k:d.get(k) for k in ('round','kernel_ms','kernel_frac_of_8TBps','round_from_boundary_ms','apply_cut_ms','waves_per_workgroup','dict_mode')})
PY
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -o bench -- \
    python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-extras > "$R/gpurun_out/prof_bench.log" 2>&1
cd "$R"
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
timeout 60 python scripts/trace_gaps.py $(dirname $(find gpurun_out/prof_bench -name "*kernel_trace.csv" | head -1)) 2>&1 | tail -30
