#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "generated" > gpurun_out/pytest_gen.log 2>&1; grep -n "passed\|failed\|Error\|error\|assert" gpurun_out/pytest_gen.log | tail -8
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error\|error" gpurun_out/pytest_gpu.log | tail -5
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_gen.json 2> gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_gen.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','load_split_ms','resolve_ms','round_from_boundary_ms','time_to_stable_cut_ms']})
print(d.get('generated_streams'))
print({k:d['roofline'].get(k) for k in ['frac','kernel_ms','traffic_over_bytes']})
PY
tail -3 gpurun_out/bench.err
