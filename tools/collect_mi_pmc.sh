#!/bin/bash
# rocprofv3 --pmc passes over the kernels of the greedy MI loop at L = 10^6 (one counter per run, kernel trace only -- never combined
# with a sys / runtime trace), folded into <out>/<prefix>_mi_pmc.json by tools/summarize_mi_pmc.py.
#   tools/collect_mi_pmc.sh [out dir] [prefix]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=${1:-gpurun_out}; P=${2:-r06}
mkdir -p "$OUT"
ITERS=1600
for C in FETCH_SIZE WRITE_SIZE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum; do
    rm -rf "$OUT/pmc_mi_$C"
    timeout 600 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_fy|k_mt|k_mi" --output-format csv -d "$OUT/pmc_mi_$C" -o pmc -- \
        python tools/bench_mi.py 1000000 256 2 0 $ITERS > /dev/null 2>&1
    cp "$OUT"/pmc_mi_$C/*counter_collection.csv "$OUT/${P}_mi_pmc_${C}_counter_collection.csv" 2> /dev/null || echo "no csv for $C"
    rm -rf "$OUT/pmc_mi_$C"
done
python tools/summarize_mi_pmc.py "$OUT" "$P" $ITERS > "$OUT/${P}_mi_pmc.json"
head -c 1500 "$OUT/${P}_mi_pmc.json"
