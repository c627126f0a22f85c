cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_cli.py -m gpu -x -q > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest.log
ACAV_FY_ECAP=64 timeout 900 python -m pytest tests/test_gpu_mi.py -m gpu -x -q > gpurun_out/r2l/pytest_ecap.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest_ecap.log
tail -4 gpurun_out/r2l/pytest.log; tail -3 gpurun_out/r2l/pytest_ecap.log
timeout 300 python tools/bench_mi_lockstep.py 100000 256 2 8 2>&1 | tail -1
timeout 300 python tools/bench_mi_lockstep.py 100000 256 2 10 2>&1 | tail -1
ACAV_FY_LEGACY=1 timeout 300 python tools/bench_mi_lockstep.py 100000 256 2 8 2>&1 | tail -1
