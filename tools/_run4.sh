cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 1000000 256 2 0 20000 > gpurun_out/r2d/mi_1m.log 2>&1
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 100000 256 2 > gpurun_out/r2d/mi_100k.log 2>&1
grep -h "acav\|us_per_iter" gpurun_out/r2d/*.log | cut -c1-300
