cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_cli.py tests/test_gpu_configs.py -m gpu -x -q -k "not cfg2 and not k1024" > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2b/ktmi -o kt -- python tools/bench_mi.py 1000000 256 2 0 3000 > gpurun_out/r2b/mi_1m.json 2>/dev/null
cp gpurun_out/r2b/ktmi/*kernel_stats.csv gpurun_out/r2b/mi_1m_kernel_stats.csv; rm -rf gpurun_out/r2b/ktmi
timeout 300 python tools/bench_mi.py 100000 256 2 > gpurun_out/r2b/mi_100k.json 2>&1
tail -15 gpurun_out/r2b/pytest.log; tail -1 gpurun_out/r2b/mi_1m.json; tail -1 gpurun_out/r2b/mi_100k.json; cut -c1-100,330-440 gpurun_out/r2b/mi_1m_kernel_stats.csv | head -12
