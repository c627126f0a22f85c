cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_cli.py tests/test_gpu_configs.py -m gpu -x -q -k "not cfg2 and not k1024" > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2m/pytest.log
tail -3 gpurun_out/r2m/pytest.log
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 1000000 256 2 0 20000 2>&1 | grep "acav\|us_per" | cut -c1-220
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 100000 256 2 2>&1 | grep "acav\|us_per" | cut -c1-220
timeout 300 python tools/bench_mi_lockstep.py 100000 256 2 10 2>&1 | tail -1 | cut -c1-400
