#!/bin/bash
# round 4, closing session on the last build: the whole GPU suite, smoke, the bench as the driver runs it, C4 shard, C5 rounds
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | tail -6
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_c3b.json 2> gpurun_out/bench.err; tail -c 300 gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c3b.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min','ms_per_step_median','time_to_stable_cut_ms','time_to_stable_cut_trials_ms','decided','cut_size')})
print({k:d['roofline'][k] for k in ('frac','kernel_ms','traffic_over_bytes','kernel_ms_filter_per_delivery')}, d['round_index']['index_build_ms'], d['cpu_baseline'])
PY
timeout 400 python scripts/c4_shard.py > gpurun_out/c4_shard.json 2> gpurun_out/c4.err; cut -c1-560 gpurun_out/c4_shard.json
timeout 400 python scripts/c5_stream.py 1000000 3 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5.err; python - <<'PY'
import json
for ln in open("gpurun_out/c5_1m.jsonl"):
    d=json.loads(ln); print({k:d.get(k) for k in ("round","kernel_ms","kernel_frac_of_8TBps","round_from_boundary_ms","apply_cut_ms")})
PY
