#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline objects and DESIGN.md's tables on the GPU box:
#   tools/collect_profiles.sh <prefix>      (e.g. r02)  -> gpurun_out/profiles/<prefix>_*
# Separate runs, as the MI355X guide prescribes: kernel trace + stats, then one --pmc pass per counter (never combined
# with a trace domain).  Copy the results into profiles/ afterwards.
set -u
P=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
stats() {  # stats <tag> <cmd...>: kernel-trace statistics of a command -> ${P}_<tag>_kernel_stats.csv, stdout -> ${P}_<tag>.txt
    local tag=$1; shift
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$tag" -o kt -- "$@" > "$OUT/${P}_${tag}.txt" 2> /dev/null
    cp "$OUT"/kt_$tag/*kernel_stats.csv "$OUT/${P}_${tag}_kernel_stats.csv" 2> /dev/null
    rm -rf "$OUT/kt_$tag"
}
pmc() {  # pmc <tag> <kernel regex> <counters> <cmd...>
    local tag=$1 re=$2 ctr=$3; shift 3
    timeout 900 rocprofv3 --pmc $ctr --kernel-include-regex "$re" --output-format csv -d "$OUT/pmc_$tag" -o pmc -- "$@" > /dev/null 2>&1
    cp "$OUT"/pmc_$tag/*counter_collection.csv "$OUT/${P}_pmc_${tag}_counter_collection.csv" 2> /dev/null
    rm -rf "$OUT/pmc_$tag"
}
# 1. the driver's command (one step: 150k greedy launches make the trace large), alone and under the tracer
timeout 900 python bench.py --steps 2 --warmup 1 > "$OUT/${P}_bench_1gpu.json" 2> /dev/null
stats bench_under_rocprof python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants
# 2. the assign sweep alone (the kernel of bench.py's `roofline`): stats, HBM traffic, pipe counters; round-1 layout for the A/B
stats assign_filter python tools/run_assign_only.py 1000000 20 filter
pmc fetch "k_assign" FETCH_SIZE python tools/run_assign_only.py 1000000 5 filter
pmc write "k_assign" WRITE_SIZE python tools/run_assign_only.py 1000000 5 filter
python tools/summarize_pmc.py "$OUT" "$P"
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES; do
    pmc "rw_$C" "k_assign_f16" $C python tools/run_assign_only.py 1000000 5 filter
done
python tools/summarize_counters.py "$OUT" "$P" > "$OUT/${P}_assign_pipe_counters.json"
stats assign_exact python tools/run_assign_only.py 1000000 3 exact
stats assign_k1024 python tools/run_assign_only.py 1000000 3 filter 1024 1024
pmc fetch_k1024 "k_assign" FETCH_SIZE python tools/run_assign_only.py 1000000 3 filter 1024 1024
pmc write_k1024 "k_assign" WRITE_SIZE python tools/run_assign_only.py 1000000 3 filter 1024 1024
python tools/summarize_pmc.py "$OUT" "$P" _k1024 1000000 1024 1024
# cfg4's widest view (d = 2048, K = 1024, the per-GPU partition of 1.25M rows): the `traffic` of bench.py --workload cfg4
pmc fetch_k1024_d2048 "k_assign" FETCH_SIZE python tools/run_assign_only.py 1250000 3 filter 2048 1024
pmc write_k1024_d2048 "k_assign" WRITE_SIZE python tools/run_assign_only.py 1250000 3 filter 2048 1024
python tools/summarize_pmc.py "$OUT" "$P" _k1024_d2048 1250000 2048 1024
# cfg4's per-GPU partition (10M clips over 8 GPUs): 1.25M rows, 2048-d visual / 128-d audio, K = 1024
(python tools/run_assign_only.py 1250000 3 filter 2048 1024; python tools/run_assign_only.py 1250000 3 filter 128 1024) > "$OUT/${P}_assign_cfg4.txt" 2> /dev/null
# 3. MI greedy: one chunk at V = 1M (3000 iterations) and V = 100k, legacy global-atomic kernels for the A/B, 8 chunks in lockstep
stats mi_1m python tools/bench_mi.py 1000000 256 2 0 3000
ACAV_FY_LEGACY=1 stats mi_1m_legacy python tools/bench_mi.py 1000000 256 2 0 3000
stats mi_100k python tools/bench_mi.py 100000 256 2
stats mi_lockstep8 python tools/bench_mi_lockstep.py 100000 256 2 8
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 1000000 256 2 0 20000 > "$OUT/${P}_mi_1m_steady.txt" 2>&1
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 100000 256 2 > "$OUT/${P}_mi_100k_steady.txt" 2>&1
# the kernels of one iteration, each ALONE on the product's code (tools/exp/build.sh fy_bench cross-compiles the harness)
if [ -x tools/exp/fy_bench ]; then
    (tools/exp/fy_bench 1000000; tools/exp/fy_bench 100000) > "$OUT/${P}_mi_kernels_alone.txt" 2>&1
fi
# 4. SGD step per shape (persistent / wide persistent / per-step launches)
: > "$OUT/${P}_train_shapes.txt"
for shape in "1024 256" "512 64" "1024 1024" "128 1024" "2048 1024" "2048 512" "2048 256"; do
    set -- $shape
    echo "d=$1 K=$2" >> "$OUT/${P}_train_shapes.txt"
    BENCH_D=$1 BENCH_K=$2 timeout 300 python tools/bench_train_b.py 32 2> /dev/null | tail -1 >> "$OUT/${P}_train_shapes.txt"
    ACAV_NO_PERSISTENT=1 BENCH_D=$1 BENCH_K=$2 timeout 300 python tools/bench_train_b.py 32 2> /dev/null | tail -1 | sed 's/^/   per-step launches: /' >> "$OUT/${P}_train_shapes.txt"
done
echo "d=1024 K=256, larger batches (the DDP path trains on global batches of 32 W rows; SURVEY 8(d): a large-batch point)" >> "$OUT/${P}_train_shapes.txt"
BENCH_D=1024 BENCH_K=256 timeout 600 python tools/bench_train_b.py 64 128 256 1024 2> /dev/null | sed 's/^/   /' >> "$OUT/${P}_train_shapes.txt"
ACAV_PROFILE_STEPS=1 BENCH_D=1024 BENCH_K=256 timeout 300 python tools/bench_train_b.py 32 > "$OUT/${P}_train_phase_cycles.txt" 2>&1
ACAV_PROFILE_STEPS=1 BENCH_D=2048 BENCH_K=1024 timeout 300 python tools/bench_train_b.py 32 > "$OUT/${P}_train_split_phase_cycles.txt" 2>&1
# 5. round 3: read ceilings of the filter's access pattern, sweep time vs re-check fraction, the out-of-core path
if [ -x tools/exp/stream_bench ]; then (cd tools/exp && ./stream_bench) > "$OUT/${P}_stream_bench.txt" 2>&1; fi
timeout 600 python tools/recheck_table.py > "$OUT/${P}_recheck_table.txt" 2>&1
timeout 900 python tools/bench_streamed.py 1000000 1024 256 2.0 > "$OUT/${P}_streamed_now.txt" 2>&1
# 6. round 4: the 8-GPU configurations' per-GPU slices (bench lines with the K = 1024 roofline against the MFMA roof), the
#    candidate-restricted re-check (kernel statistics + counters at 62 % undecided rows)
timeout 900 python bench.py --workload cfg4 --steps 2 --warmup 1 > "$OUT/${P}_cfg4_slice.json" 2> /dev/null
timeout 1800 python bench.py --workload cfg5 --steps 1 --warmup 1 > "$OUT/${P}_cfg5_slice.json" 2> /dev/null
stats recheck_hard python tools/recheck_table.py 1000000 1024 0.06,0.05
: > "$OUT/${P}_cand_pmc.txt"
for C in "TCC_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    tag=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $C --kernel-include-regex "k_assign_cand|k_assign_f16_rw" --output-format csv -d "$OUT/pmc_cand_$tag" -o pmc -- python tools/recheck_table.py 1000000 1024 0.05 > /dev/null 2>&1
    python - "$OUT/pmc_cand_$tag" >> "$OUT/${P}_cand_pmc.txt" <<'PY'
import csv, glob, collections, re, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Kernel_Name"])
        agg[(k.group(0) if k else "?", r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-60s %-34s launches %3d  mean per launch %.0f" % (k, c, len(v), sum(v) / len(v)))
PY
    rm -rf "$OUT/pmc_cand_$tag"
done
# 7. round 5: the real pipeline's shape (ten clusterings, P = 45) as bench lines, ten clusterings alone vs in one call, the driver's
#    command with --verify, the cfg4 pair and the K = d = 1024 pair side by side
timeout 900 python bench.py --workload real10 --steps 2 --warmup 1 > "$OUT/${P}_real10_k32.json" 2> /dev/null
timeout 900 python bench.py --workload real10 --k 256 --steps 2 --warmup 1 > "$OUT/${P}_real10_k256.json" 2> /dev/null
(BENCH_K=32 python tools/bench_train_real10.py; BENCH_K=256 python tools/bench_train_real10.py) > "$OUT/${P}_train_real10.txt" 2>&1
timeout 600 python bench.py --steps 2 --warmup 1 --verify --no-cpu-baseline --no-variants > "$OUT/${P}_bench_verify.json" 2> /dev/null
(python tools/bench_train_multi.py; BENCH_D=2048 BENCH_D2=128 python tools/bench_train_multi.py; BENCH_K=256 python tools/bench_train_multi.py) > "$OUT/${P}_train_multi.txt" 2>&1
for f in "$OUT"/${P}_*.txt; do sed -i '/amdgpu.ids/d' "$f"; done
for f in "$OUT/${P}_mi_1m.txt" "$OUT/${P}_mi_1m_legacy.txt" "$OUT/${P}_mi_100k.txt" "$OUT/${P}_mi_lockstep8.txt" "$OUT/${P}_bench_under_rocprof.txt"; do grep "^{" "$f" | tail -1 > "$f.tmp" && mv "$f.tmp" "${f%.txt}.json" && rm -f "$f"; done
ls -la "$OUT"
