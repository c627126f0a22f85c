#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline object on the GPU box:
#   tools/collect_profiles.sh <prefix>      (e.g. r01_final)  -> gpurun_out/profiles/<prefix>_*
# Three separate runs of the same command, as the MI355X guide prescribes: kernel trace + stats, then one --pmc
# pass per counter (never combined with a trace domain).  Copy the results into profiles/ afterwards.
set -u
P=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 $CMD > "$OUT/${P}_bench_1gpu.json" 2> /dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $CMD > "$OUT/${P}_bench_under_rocprof.json" 2> /dev/null
cp "$OUT"/kt/*kernel_stats.csv "$OUT/${P}_kernel_stats.csv" 2> /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --kernel-include-regex "k_assign" --output-format csv -d "$OUT/pmc_$C" -o pmc -- $CMD > /dev/null 2>&1
    L=$(echo $C | tr 'A-Z' 'a-z' | sed 's/_size//')
    cp "$OUT"/pmc_$C/*counter_collection.csv "$OUT/${P}_pmc_${L}_counter_collection.csv" 2> /dev/null
done
python tools/summarize_pmc.py "$OUT" "$P"
# the other kernels DESIGN.md quotes: MI greedy (one chunk, 8 chunks in lockstep) and the exact fp32 assign sweep
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktmi" -o kt -- python tools/bench_mi.py 100000 256 2 > "$OUT/${P}_mi_one_chunk.json" 2> /dev/null
cp "$OUT"/ktmi/*kernel_stats.csv "$OUT/${P}_mi_one_chunk_kernel_stats.csv" 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktml" -o kt -- python tools/bench_mi_lockstep.py 100000 256 2 8 > "$OUT/${P}_mi_lockstep8.json" 2> /dev/null
cp "$OUT"/ktml/*kernel_stats.csv "$OUT/${P}_mi_lockstep8_kernel_stats.csv" 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktex" -o kt -- python tools/run_assign_only.py 1000000 3 exact > "$OUT/${P}_assign_exact.txt" 2> /dev/null
cp "$OUT"/ktex/*kernel_stats.csv "$OUT/${P}_assign_exact_kernel_stats.csv" 2> /dev/null
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "k_assign_f32" --output-format csv -d "$OUT/pmcex" -o pmc -- python tools/run_assign_only.py 1000000 2 exact > /dev/null 2>&1
cp "$OUT"/pmcex/*counter_collection.csv "$OUT/${P}_assign_exact_pmc_mfma_counter_collection.csv" 2> /dev/null
for f in "$OUT/${P}_mi_one_chunk.json" "$OUT/${P}_mi_lockstep8.json"; do grep "^{" "$f" | tail -1 > "$f.tmp" && mv "$f.tmp" "$f"; done
rm -rf "$OUT/kt" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT/ktmi" "$OUT/ktml" "$OUT/ktex" "$OUT/pmcex"
ls -la "$OUT"
