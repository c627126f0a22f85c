cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_contrastive.py -m gpu -x -q > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest.log
tail -40 gpurun_out/r2k/pytest.log
