#!/bin/bash
# experiment harnesses around the product kernels (not shipped, not used by the product)
# usage: tools/exp/build.sh <output name> [-D...]
cd "$(dirname "$0")/../.."
out=${1:-fy_bench}; shift
F="-DACAV_EXPERIMENT_BUILD --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-inline-asm -I include -I acav100m_amd/csrc"
hipcc $F "$@" -o tools/exp/$out tools/exp/fy_bench.hip acav100m_amd/csrc/acav_common.hip acav100m_amd/csrc/acav_mtjump.hip
