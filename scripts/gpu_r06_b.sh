#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r06b; mkdir -p $O
filter() { grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP\|amdgpu.ids"; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | filter | tail -2
timeout 600 python scripts/step_ab_settle.py 40 2>&1 | filter | tee $O/step_ab_settle.txt | tail -4
timeout 1700 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; filter < $O/pytest_gpu.log | tail -12
