#!/bin/bash
# experiment harness around the product's K <= 256 assign filter under sustained load (not shipped):
#   tools/exp/build_sustained.sh <output name> [-DACAV_ABL_NOMFMA ...]
cd "$(dirname "$0")/../.."
out=${1:-sustained_bench}; shift
F="-DACAV_EXPERIMENT_BUILD --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-inline-asm -I include -I acav100m_amd/csrc"
hipcc $F "$@" -o tools/exp/$out tools/exp/sustained_bench.hip acav100m_amd/csrc/acav_common.hip
