cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
timeout 600 python bench.py --steps 1 --warmup 1 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2a/ktmi -o kt -- python tools/bench_mi.py 1000000 256 2 0 800 > gpurun_out/r2a/mi_1m.json 2>/dev/null
cp gpurun_out/r2a/ktmi/*kernel_stats.csv gpurun_out/r2a/mi_1m_kernel_stats.csv; rm -rf gpurun_out/r2a/ktmi
tail -5 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/bench.json | cut -c1-1500; cat gpurun_out/r2a/mi_1m_kernel_stats.csv | head -12
