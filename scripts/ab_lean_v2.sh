#!/bin/bash
# A/B of the prepared tally-kernel variants (compile-time switches of csrc/tally_kernel.h) against the default kernel:
#   v2     -DRAPID_LEAN_V2=1        the lean window with fewer instructions
#   hint   -DRAPID_CAREFUL_HINT=1   careful path: next attempt sized from the crossing mask, end-game rule
#   early  -DRAPID_EARLY_CERT=1     lean path: bound certificate while there is no witness yet
#   all    the three together
#   fast   -DRAPID_FAST_WINDOW=1    windows applied without return values once a witness exists; flat pair pass for the owed reports
#   all4   the four together
#   pairs  -DRAPID_DMA_PAIRS=1      stream topped up two KiB per loop trip, running LDS address
#   all5   the five together
# On the GPU box: parity tests on the `all`, `all4` and `all5` libraries, then the tally kernel time of all nine, interleaved
# (box noise ~5 %).
# Build the variants first (works here or on the box):  bash scripts/ab_lean_v2.sh build
set -u
cd "$(dirname "$0")/.."
declare -A DEFS=( [v2]="-DRAPID_LEAN_V2=1" [hint]="-DRAPID_CAREFUL_HINT=1" [early]="-DRAPID_EARLY_CERT=1"
                  [all]="-DRAPID_LEAN_V2=1 -DRAPID_CAREFUL_HINT=1 -DRAPID_EARLY_CERT=1" [fast]="-DRAPID_FAST_WINDOW=1"
                  [all4]="-DRAPID_LEAN_V2=1 -DRAPID_CAREFUL_HINT=1 -DRAPID_EARLY_CERT=1 -DRAPID_FAST_WINDOW=1" [pairs]="-DRAPID_DMA_PAIRS=1"
                  [all5]="-DRAPID_LEAN_V2=1 -DRAPID_CAREFUL_HINT=1 -DRAPID_EARLY_CERT=1 -DRAPID_FAST_WINDOW=1 -DRAPID_DMA_PAIRS=1"
                  # with cheap windows fewer waves are needed: a deeper ring (15 KiB = 768 records: 10 KiB in flight per wave, 7 waves per CU)
                  [all5r15]="-DRAPID_LEAN_V2=1 -DRAPID_CAREFUL_HINT=1 -DRAPID_EARLY_CERT=1 -DRAPID_FAST_WINDOW=1 -DRAPID_DMA_PAIRS=1 -DRAPID_RING_SLOTS=15" )
for v in v2 hint early all fast all4 pairs all5 all5r15; do
    lib="$PWD/rapid_amd/librapid_mi355x_$v.so"
    if [ "${1:-}" = "build" ] || [ ! -f "$lib" ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC ${DEFS[$v]} -Irapid_amd/csrc \
            rapid_amd/csrc/engine.hip rapid_amd/csrc/host_abi.cpp -o "$lib" -lrccl || exit 1
    fi
done
[ "${1:-}" = "build" ] && exit 0
mkdir -p gpurun_out
for v in all all4 all5; do
    RAPID_MI355X_LIB="$PWD/rapid_amd/librapid_mi355x_$v.so" timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_$v.log 2>&1
    echo -n "$v: "; tail -1 gpurun_out/pytest_gpu_$v.log
done
timeout 1200 python scripts/ab_variants.py C3b 3 20 2>&1 | tee gpurun_out/ab_variants.log | tail -40
