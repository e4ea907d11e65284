#!/bin/bash
# Builds the profiling variant of the library (phase timers in the tally kernel) next to the product one and runs it.
set -eu
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DRAPID_TEST_BUILD -DRAPID_MEASUREMENT_BUILD -DRAPID_PHASE_TIMERS -Irapid_amd/csrc rapid_amd/csrc/engine.hip rapid_amd/csrc/host_abi.cpp \
    -o rapid_amd/librapid_mi355x_timers.so -lrccl
RAPID_MI355X_LIB="$PWD/rapid_amd/librapid_mi355x_timers.so" python scripts/phase_timers.py "${1:-C3b}"
