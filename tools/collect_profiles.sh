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
rm -rf "$OUT/kt" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
ls -la "$OUT"
