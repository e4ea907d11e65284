#!/bin/bash
# Builds measurement variants of the library next to the product one: scripts/build_variants.sh tag "-DX=1 -DY=2" [tag "..."] ...
# Every variant is a TEST build with the measurement hooks of csrc/tally_probes.inc switched on (-DRAPID_MEASUREMENT_BUILD):
#   scripts/build_variants.sh timers "-DRAPID_PHASE_TIMERS" stamps "-DRAPID_PHASE_TIMERS -DRAPID_BLOCK_STAMPS" sets3 "-DRAPID_SETS_BOUNDARY=3"
set -eu
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DRAPID_TEST_BUILD -DRAPID_MEASUREMENT_BUILD $defs -Irapid_amd/csrc rapid_amd/csrc/engine.hip rapid_amd/csrc/host_abi.cpp \
      -o rapid_amd/librapid_mi355x_$tag.so -lrccl &
done
wait
ls -la rapid_amd/librapid_mi355x_*.so
