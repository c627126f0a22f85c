#!/bin/bash
# experiment harness around the product's assign filter (not shipped): tools/exp/build_assign.sh <output name> [-D...]
cd "$(dirname "$0")/../.."
out=${1:-assign_bench}; shift
F="-DACAV_EXPERIMENT_BUILD --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-inline-asm -I include -I acav100m_amd/csrc"
hipcc $F "$@" -o tools/exp/$out tools/exp/assign_bench.hip acav100m_amd/csrc/acav_common.hip
