cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_configs.py -m gpu -x -q -k "not cfg3 and not cfg1 and not cfg2" > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j/pytest.log
tail -5 gpurun_out/r2j/pytest.log
for shape in "1024 1024" "128 1024" "2048 1024" "1024 256" "512 64"; do set -- $shape; echo "d=$1 K=$2"; BENCH_D=$1 BENCH_K=$2 timeout 300 python tools/bench_train_b.py 32 2>&1 | tail -1; ACAV_NO_PERSISTENT=1 BENCH_D=$1 BENCH_K=$2 timeout 300 python tools/bench_train_b.py 32 2>&1 | tail -1 | sed 's/^/   per-step path: /'; done
