#!/bin/bash
# Round-2 profile session: rocprofv3 kernel statistics of the bench command, FETCH_SIZE / WRITE_SIZE passes of the tally
# kernel (calibrated on the stream probe of the same run), SQ counters.  Summaries are copied to profiles/ by hand.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_bench" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch" -o pmc -- \
    python "$GRAFT_REPO_ROOT/scripts/prof_tally.py" C3b 3 > "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_write" -o pmc -- \
    python "$GRAFT_REPO_ROOT/scripts/prof_tally.py" C3b 3 > "$GRAFT_REPO_ROOT/gpurun_out/prof_write.log" 2>&1
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_$tag" -o pmc -- \
        python "$GRAFT_REPO_ROOT/scripts/prof_tally.py" C3b 3 > "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_$tag.log" 2>&1
done
cd "$GRAFT_REPO_ROOT"
grep -h "^workload" gpurun_out/prof_fetch.log | tail -1
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -2
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-170
