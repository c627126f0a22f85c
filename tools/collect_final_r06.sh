#!/bin/bash
# Round 6's final collection on the GPU box (the subset of tools/collect_profiles.sh that the round's numbers are quoted from; every
# command under `timeout`): -> gpurun_out/profiles/r06_*; copy into profiles/ afterwards.
set -u
P=r06
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
stats() { local tag=$1; shift
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$tag" -o kt -- "$@" > "$OUT/${P}_${tag}.txt" 2> /dev/null
    cp "$OUT"/kt_$tag/*kernel_stats.csv "$OUT/${P}_${tag}_kernel_stats.csv" 2> /dev/null
    rm -rf "$OUT/kt_$tag"; }
pmc() { local tag=$1 re=$2 ctr=$3; shift 3
    timeout 900 rocprofv3 --pmc $ctr --kernel-include-regex "$re" --output-format csv -d "$OUT/pmc_$tag" -o pmc -- "$@" > /dev/null 2>&1
    cp "$OUT"/pmc_$tag/*counter_collection.csv "$OUT/${P}_pmc_${tag}_counter_collection.csv" 2> /dev/null
    rm -rf "$OUT/pmc_$tag"; }
timeout 900 python bench.py --steps 2 --warmup 1 > "$OUT/${P}_bench_1gpu.json" 2> /dev/null
stats bench_under_rocprof python bench.py --steps 1 --warmup 0 --no-cpu-baseline
stats assign_filter python tools/run_assign_only.py 1000000 200 filter
pmc fetch "k_assign" FETCH_SIZE python tools/run_assign_only.py 1000000 5 filter
pmc write "k_assign" WRITE_SIZE python tools/run_assign_only.py 1000000 5 filter
python tools/summarize_pmc.py "$OUT" "$P"
stats assign_k1024 python tools/run_assign_only.py 1000000 3 filter 1024 1024
timeout 900 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${P}_cfg4_slice.json" 2> /dev/null
timeout 1800 python bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/${P}_cfg5_slice.json" 2> /dev/null
timeout 900 python bench.py --workload real10 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${P}_real10_k32.json" 2> /dev/null
timeout 900 python bench.py --workload real10 --k 256 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${P}_real10_k256.json" 2> /dev/null
stats mi_1m python tools/bench_mi.py 1000000 256 2 0 3000
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 1000000 256 2 0 20000 > "$OUT/${P}_mi_1m_steady.txt" 2>&1
: > "$OUT/${P}_train_shapes.txt"
for shape in "1024 256" "512 64" "1024 1024" "768 1024" "128 1024" "2048 1024" "2048 512" "2048 256" "1408 256" "2304 256"; do
    set -- $shape
    echo "d=$1 K=$2" >> "$OUT/${P}_train_shapes.txt"
    BENCH_D=$1 BENCH_K=$2 timeout 300 python tools/bench_train_b.py 32 2> /dev/null | tail -1 >> "$OUT/${P}_train_shapes.txt"
    ACAV_NO_PERSISTENT=1 BENCH_D=$1 BENCH_K=$2 timeout 300 python tools/bench_train_b.py 32 2> /dev/null | tail -1 | sed 's/^/   per-step launches: /' >> "$OUT/${P}_train_shapes.txt"
done
ACAV_PROFILE_STEPS=1 BENCH_D=1024 BENCH_K=256 timeout 300 python tools/bench_train_b.py 32 > "$OUT/${P}_train_phase_cycles.txt" 2>&1
(timeout 300 python tools/bench_train_multi.py; BENCH_D=2048 BENCH_D2=128 timeout 300 python tools/bench_train_multi.py; BENCH_K=256 timeout 300 python tools/bench_train_multi.py) > "$OUT/${P}_train_multi.txt" 2>&1
(BENCH_K=32 timeout 600 python tools/bench_train_real10.py; BENCH_K=256 timeout 600 python tools/bench_train_real10.py) > "$OUT/${P}_train_real10.txt" 2>&1
for f in "$OUT"/${P}_*.txt; do sed -i '/amdgpu.ids/d' "$f"; done
for f in "$OUT/${P}_mi_1m.txt" "$OUT/${P}_bench_under_rocprof.txt"; do grep "^{" "$f" | tail -1 > "$f.tmp" && mv "$f.tmp" "${f%.txt}.json" && rm -f "$f"; done
ls -la "$OUT"
