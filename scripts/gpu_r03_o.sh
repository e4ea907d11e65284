#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now ', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
RAPID_MI355X_LIB=$PWD/rapid_amd/librapid_mi355x_b45.so timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b45 ', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now ', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
