#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error\|error" gpurun_out/pytest_gpu.log | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 300 gpurun_out/bench_c3b.json | head -c 200; echo
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -o bench -- \
    python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > "$R/gpurun_out/prof_bench.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/prof_$c" -o pmc -- python "$R/scripts/prof_tally.py" C3b 3 > "$R/gpurun_out/prof_$c.log" 2>&1
done
cd "$R"
bash scripts/pmc_sq.sh > gpurun_out/sq_summary.txt 2>&1; tail -22 gpurun_out/sq_summary.txt | head -21
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-150
