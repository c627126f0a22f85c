#!/bin/bash
# experiment builds of acav_kmeans_assign.hip linked against the product's other objects:
#   tools/exp/build_assign_variant.sh <name> [-D...]  ->  tools/exp/libacav_hip_<name>.so   (ACAV_LIB_PATH)
cd "$(dirname "$0")/../.."
name=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -Wno-unused-function -Wno-inline-asm -I include"
hipcc $F -DACAV_EXPERIMENT_BUILD "$@" -c acav100m_amd/csrc/acav_kmeans_assign.hip -o build/obj_exp_assign_$name.o || exit 1
objs=$(ls build/obj/*.o | grep -v "acav_kmeans_assign.o\|acav_mi_empty.o")
hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o tools/exp/libacav_hip_$name.so $objs build/obj_exp_assign_$name.o
