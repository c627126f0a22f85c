#!/bin/bash
# experiment: the product library with every kernel of the MI greedy loop returning at once (-DACAV_MI_ABL_EMPTY) ->
# tools/exp/libacav_hip_emptymi.so.  Run the greedy loop on it with ACAV_MI_TIMING=1 to see the HOST cost per iteration of the
# same launch pattern against an idle GPU:
#   ACAV_LIB_PATH=tools/exp/libacav_hip_emptymi.so ACAV_MI_TIMING=1 python tools/bench_mi.py 1000000 256 2 0 20000
# (needs build/obj/*.o of a normal build: python __graft_entry__.py)
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -Wno-unused-function -Wno-inline-asm -I include"
hipcc $F -DACAV_EXPERIMENT_BUILD -DACAV_MI_ABL_EMPTY -c acav100m_amd/csrc/acav_mi.hip -o build/obj/acav_mi_empty.o || exit 1
objs=$(ls build/obj/*.o | grep -v "acav_mi.o\|acav_mi_empty.o")
hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o tools/exp/libacav_hip_emptymi.so $objs build/obj/acav_mi_empty.o
