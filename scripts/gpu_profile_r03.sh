#!/bin/bash
# Round-3 profile session: parity tests, bench, rocprofv3 kernel statistics of the bench command, FETCH_SIZE / WRITE_SIZE
# passes of the tally kernel (calibrated on the stream probe of the same run), SQ counters, phase timers, C4 shard, C5 rounds.
# Summaries are copied to profiles/ by hand.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error\|error" gpurun_out/pytest_gpu.log | tail -5
timeout 600 python bench.py > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 600 gpurun_out/bench_c3b.json | head -c 300; echo
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -o bench -- \
    python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > "$R/gpurun_out/prof_bench.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/prof_$c" -o pmc -- python "$R/scripts/prof_tally.py" C3b 3 > "$R/gpurun_out/prof_$c.log" 2>&1
done
cd "$R"
bash scripts/pmc_sq.sh > gpurun_out/sq_summary.txt 2>&1; tail -22 gpurun_out/sq_summary.txt
timeout 600 bash scripts/phase_timers.sh C3b > gpurun_out/timers.log 2>&1; grep -v "^receivers  " gpurun_out/timers.log | tail -12
rm -f rapid_amd/librapid_mi355x_timers.so
timeout 300 python scripts/c4_shard.py > gpurun_out/c4_shard.json 2>/dev/null; tail -1 gpurun_out/c4_shard.json | cut -c1-400
timeout 600 python scripts/c5_stream.py 1000000 3 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5_1m.err; cut -c1-300 gpurun_out/c5_1m.jsonl
timeout 300 python bench.py --config C2 --no-pmc > gpurun_out/bench_c2.json 2>/dev/null
timeout 300 python bench.py --config C3a --no-pmc > gpurun_out/bench_c3a.json 2>/dev/null
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-170
grep -h "^workload" gpurun_out/prof_FETCH_SIZE.log | tail -1 | cut -c1-300
