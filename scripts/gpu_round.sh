#!/bin/bash
# One GPU session: parity tests, smoke, bench, rocprofv3 kernel statistics of the bench command, PMC traffic pass.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 2500 gpurun_out/bench.json
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_bench" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch" -o pmc -- \
    python "$GRAFT_REPO_ROOT/scripts/prof_tally.py" C3b 3 > "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_write" -o pmc -- \
    python "$GRAFT_REPO_ROOT/scripts/prof_tally.py" C3b 3 > "$GRAFT_REPO_ROOT/gpurun_out/prof_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
grep -h "^workload" gpurun_out/prof_fetch.log | tail -1
head -5 gpurun_out/prof_bench/bench_kernel_stats.csv | cut -c1-160
