cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_cli.py -m gpu -x -q > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2i/pytest.log
tail -30 gpurun_out/r2i/pytest.log
