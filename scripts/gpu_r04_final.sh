#!/bin/bash
# round 4, final session: the whole GPU suite, smoke, the bench as the driver runs it, rocprofv3 kernel statistics + time line of the
# bench command, SQ counter passes, phase timers, rounds probe, C4 shard, C3a / C2 bench lines
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | tail -6
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 300 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bench_c3b.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','ms_per_step_median','time_to_stable_cut_ms','decided','cut_size')})
    print({k:d['roofline'][k] for k in ('frac','kernel_ms','traffic_over_bytes','kernel_ms_filter_per_delivery')}, d['round_index']['index_build_ms'], d['cpu_baseline'])
except Exception as e:
    print("bench line unreadable", e)
PY
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -o bench -- \
    python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-extras > "$R/gpurun_out/prof_bench.log" 2>&1
cd "$R"
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
timeout 60 python scripts/trace_gaps.py $(dirname $(find gpurun_out/prof_bench -name "*kernel_trace.csv" | head -1)) > gpurun_out/trace_gaps.txt 2>&1; tail -12 gpurun_out/trace_gaps.txt
bash scripts/pmc_sq.sh > gpurun_out/sq_summary.txt 2>&1; tail -24 gpurun_out/sq_summary.txt
RAPID_MI355X_LIB=$PWD/rapid_amd/librapid_mi355x_timers.so timeout 200 python scripts/phase_timers.py C3b > gpurun_out/timers.log 2>&1; head -9 gpurun_out/timers.log
timeout 300 python scripts/rounds_probe.py > gpurun_out/rounds_probe.txt 2>&1; tail -8 gpurun_out/rounds_probe.txt
timeout 400 python scripts/c4_shard.py > gpurun_out/c4_shard.json 2> gpurun_out/c4.err; cut -c1-600 gpurun_out/c4_shard.json
timeout 300 python bench.py --config C3a --no-cpu-baseline --no-pmc --no-extras > gpurun_out/bench_c3a.json 2>/dev/null; cut -c1-200 gpurun_out/bench_c3a.json
timeout 300 python bench.py --config C2 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/bench_c2.json 2>/dev/null; cut -c1-200 gpurun_out/bench_c2.json
