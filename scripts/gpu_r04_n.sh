#!/bin/bash
# round 4, session N: answers published with one store instruction / ordinary stores + one fence, index timing events on request only,
# per-slot tables through typed LDS pointers -- parity subset, one whole round under the knobs, the index by events, bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "not million and not 100k" > gpurun_out/pytest_gpu_n.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" gpurun_out/pytest_gpu_n.log | tail -4
timeout 300 python scripts/step_ab.py C3b 50 > gpurun_out/step_ab.log 2>&1; grep "knob      0\|knob 131072" gpurun_out/step_ab.log | tail -6
timeout 200 python scripts/index_probe.py > gpurun_out/index_probe.log 2>&1; tail -4 gpurun_out/index_probe.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_n.json 2> gpurun_out/bench_n.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_n.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_min", "time_to_stable_cut_ms")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["round_index"]["index_build_ms"])
PY
