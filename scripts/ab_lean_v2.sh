#!/bin/bash
# A/B of the RAPID_LEAN_V2 lean window (tally_kernel.h) against the default kernel, on the GPU box:
#   parity tests on the variant library, then the tally kernel time of both, interleaved (box noise is ~5 %).
# Build the variant first (works here or on the box):  bash scripts/ab_lean_v2.sh build
set -u
cd "$(dirname "$0")/.."
V2="$PWD/rapid_amd/librapid_mi355x_v2.so"
if [ "${1:-}" = "build" ] || [ ! -f "$V2" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DRAPID_LEAN_V2=1 -Irapid_amd/csrc rapid_amd/csrc/engine.hip rapid_amd/csrc/host_abi.cpp \
        -o "$V2" -lrccl || exit 1
    [ "${1:-}" = "build" ] && exit 0
fi
mkdir -p gpurun_out
RAPID_MI355X_LIB="$V2" timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_v2.log 2>&1; tail -2 gpurun_out/pytest_gpu_v2.log
for i in 1 2 3; do
    for lib in default v2; do
        if [ "$lib" = v2 ]; then export RAPID_MI355X_LIB="$V2"; else unset RAPID_MI355X_LIB; fi
        echo -n "$lib run $i: "
        timeout 300 python scripts/prof_tally.py C3b 20 2>&1 | grep -h "^workload\|tally" | tail -1
    done
done
unset RAPID_MI355X_LIB
