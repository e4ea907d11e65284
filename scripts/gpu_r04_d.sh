#!/bin/bash
# round 4, session D: GPU suite, smoke, bench (C3b) with PMC traffic, rocprofv3 kernel stats of the bench command, C4 shard, C5 rounds at 10^6
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 600 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_c3b.json'))
    print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','time_to_stable_cut_ms','decided','cut_size','load_from_host_ms')})
    print(d['roofline'])
    print(d['generated_streams'])
    print(d.get('cpu_baseline'))
except Exception as e:
    print("bench line unreadable", e)
PY
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -o bench -- \
    python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > "$R/gpurun_out/prof_bench.log" 2>&1
cd "$R"
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-170
timeout 600 python scripts/c4_shard.py 8 10 > gpurun_out/c4_shard.json 2> gpurun_out/c4.err; tail -c 300 gpurun_out/c4.err; cat gpurun_out/c4_shard.json | cut -c1-1500
timeout 900 python scripts/c5_stream.py 1000000 3 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5.err; tail -c 300 gpurun_out/c5.err; cut -c1-900 gpurun_out/c5_1m.jsonl
