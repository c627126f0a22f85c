# refresh of the MI-related evidence after the sorted-run appends of k_fy_part (round 3, late)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=r03
OUT=gpurun_out/profiles; mkdir -p $OUT
stats() { local tag=$1; shift
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$tag" -o kt -- "$@" > "$OUT/${P}_${tag}.txt" 2> /dev/null
    cp "$OUT"/kt_$tag/*kernel_stats.csv "$OUT/${P}_${tag}_kernel_stats.csv" 2> /dev/null; rm -rf "$OUT/kt_$tag"; }
timeout 900 python bench.py --steps 2 --warmup 1 2>/dev/null | tail -1 > "$OUT/${P}_bench_final.json"
stats bench_under_rocprof python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants
stats mi_1m python tools/bench_mi.py 1000000 256 2 0 3000
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 1000000 256 2 0 20000 > "$OUT/${P}_mi_1m_steady.txt" 2>&1
(tools/exp/fy_bench 1000000; tools/exp/fy_bench 100000) > "$OUT/${P}_mi_kernels_alone.txt" 2>&1
for f in "$OUT"/${P}_*.txt; do sed -i '/amdgpu.ids/d' "$f"; done
for f in "$OUT/${P}_mi_1m.txt" "$OUT/${P}_bench_under_rocprof.txt"; do grep "^{" "$f" | tail -1 > "$f.tmp" && mv "$f.tmp" "${f%.txt}.json" && rm -f "$f"; done
mkdir -p gpurun_out/pmc_mi
for C in "TCC_REQ_sum TCC_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --kernel-include-regex "k_fy_part" --output-format csv -d gpurun_out/pmc_mi/$tag -o pmc -- python tools/bench_mi.py 1000000 256 2 0 512 > /dev/null 2>&1
done
python - <<'PY'
import csv,glob,collections,json,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_mi/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:{"launches":len(v),"mean_per_launch":sum(v)/len(v)} for c,v in d.items()} for k,d in agg.items()}
json.dump(out,open("gpurun_out/profiles/r03_mi_pmc_part_sorted_runs.json","w"),indent=1)
print(out)
PY
rm -rf gpurun_out/pmc_mi
ls -la $OUT
