#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error\|error" gpurun_out/pytest_gpu.log | tail -8
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','ms_per_step_min','ms_per_step_without_index','load_split_ms','resolve_ms','round_from_boundary_ms','time_to_stable_cut_ms','n_ranks_seen']})
r=d['roofline']; print({k:r.get(k) for k in ['achieved','frac','traffic','traffic_over_bytes','kernel_ms','kernel_ms_filter_per_delivery','stream_probe_gbs']}); print(d['round_index'])
PY
tail -2 gpurun_out/bench.err
timeout 600 python scripts/c5_stream.py 1000000 2 1024 > gpurun_out/c5_1m.jsonl 2> gpurun_out/c5_1m.err; cat gpurun_out/c5_1m.jsonl | cut -c1-620; tail -2 gpurun_out/c5_1m.err
timeout 300 python scripts/c4_shard.py > gpurun_out/c4.log 2>&1; tail -1 gpurun_out/c4.log | cut -c1-700
