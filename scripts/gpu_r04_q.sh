#!/bin/bash
# round 4, session Q: the hot adjacency of large populations slot by slot in parallel (index_edges_kernel), the vote verification with
# a wave per receiver for large bitmaps -- parity at 10^5 / 10^6 nodes and of the vote paths, C5 rounds, one C5 round on the time line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "100k or million or c4_ or memory_mode or vote or round_decides or two_ranks" > gpurun_out/pytest_gpu_q.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" gpurun_out/pytest_gpu_q.log | tail -3
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf $R/gpurun_out/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -o c5 -- python $R/scripts/c5_stream.py 1000000 2 1024 > $R/gpurun_out/c5_traced.jsonl 2> $R/gpurun_out/prof_c5.log
cd $R
grep -h "index_\|vote_verify\|dict_entries" gpurun_out/prof_c5/c5_kernel_stats.csv | cut -c1-60,200-400 | head; cut -c1-330 gpurun_out/c5_traced.jsonl
timeout 400 python scripts/c5_stream.py 1000000 3 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5.err; python - <<'PY'
import json
for ln in open("gpurun_out/c5_1m.jsonl"):
    d=json.loads(ln); print({k:d.get(k) for k in ("round","kernel_ms","kernel_frac_of_8TBps","round_from_boundary_ms","apply_cut_ms")})
PY
