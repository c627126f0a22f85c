set -u
P=r04
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
stats() { local tag=$1; shift
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$tag" -o kt -- "$@" > "$OUT/${P}_${tag}.txt" 2> /dev/null
    cp "$OUT"/kt_$tag/*kernel_stats.csv "$OUT/${P}_${tag}_kernel_stats.csv" 2> /dev/null
    rm -rf "$OUT/kt_$tag"; }
timeout 900 python bench.py --steps 2 --warmup 1 > "$OUT/${P}_bench_1gpu_s3.json" 2> /dev/null
stats bench_under_rocprof python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants
timeout 900 python bench.py --workload cfg4 --steps 2 --warmup 1 > "$OUT/${P}_cfg4_slice.json" 2> /dev/null
timeout 1800 python bench.py --workload cfg5 --steps 1 --warmup 1 > "$OUT/${P}_cfg5_slice.json" 2> /dev/null
stats mi_1m python tools/bench_mi.py 1000000 256 2 0 3000
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 1000000 256 2 0 20000 > "$OUT/${P}_mi_1m_steady.txt" 2>&1
if [ -x tools/exp/fy_bench ]; then (tools/exp/fy_bench 1000000; tools/exp/fy_bench 100000) > "$OUT/${P}_mi_kernels_alone.txt" 2>&1; fi
: > "$OUT/${P}_train_multi.txt"
for s in "1024 1024 256" "1024 1024 512" "1024 1024 1024" "896 896 1024" "128 128 1024" "2048 128 1024" "1024 128 1024"; do set -- $s
    BENCH_D=$1 BENCH_D2=$2 BENCH_K=$3 timeout 300 python tools/bench_train_multi.py 2> /dev/null | tail -1 >> "$OUT/${P}_train_multi.txt"
done
ACAV_WIDE_NRP=2 BENCH_D=1024 BENCH_K=1024 timeout 300 python tools/bench_train_multi.py 2> /dev/null | tail -1 | sed 's/^/ACAV_WIDE_NRP=2 (the two-row-pass form forced for the single handle too): /' >> "$OUT/${P}_train_multi.txt"
tools/exp/colds_probe_bench 135744 240 > "$OUT/${P}_colds_probe.txt" 2>&1
for f in "$OUT"/${P}_*.txt; do sed -i '/amdgpu.ids/d' "$f"; done
for f in "$OUT/${P}_mi_1m.txt" "$OUT/${P}_bench_under_rocprof.txt"; do grep "^{" "$f" | tail -1 > "$f.tmp" && mv "$f.tmp" "${f%.txt}.json" && rm -f "$f"; done
ls -la "$OUT"
# SGD step per shape after the compact update
: > "$OUT/${P}_train_shapes_compact.txt"
for shape in "1024 256" "512 64" "1024 1024" "128 1024" "2048 1024" "2048 512" "2048 256" "1408 256" "2304 256" "1408 1024"; do
    set -- $shape
    echo -n "d=$1 K=$2: " >> "$OUT/${P}_train_shapes_compact.txt"
    BENCH_D=$1 BENCH_K=$2 timeout 300 python tools/bench_train_b.py 32 2> /dev/null | tail -1 >> "$OUT/${P}_train_shapes_compact.txt"
done
ACAV_PROFILE_STEPS=1 BENCH_D=1024 BENCH_K=256 timeout 300 python tools/bench_train_b.py 32 2>&1 | grep -a "acav\|us/step" > "$OUT/${P}_train_phase_cycles_compact.txt"
