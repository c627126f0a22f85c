cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_configs.py -m gpu -x -q -k "not cfg3 and not cfg1" > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log
timeout 300 python tools/run_assign_only.py 1000000 20 filter > gpurun_out/r2f/assign_rw256.txt 2>&1
ACAV_FILTER_V2=1 timeout 300 python tools/run_assign_only.py 1000000 20 filter > gpurun_out/r2f/assign_rw128.txt 2>&1
ACAV_FILTER_V1=1 timeout 300 python tools/run_assign_only.py 1000000 20 filter > gpurun_out/r2f/assign_v1.txt 2>&1
tail -4 gpurun_out/r2f/pytest.log; for f in rw256 rw128 v1; do echo $f; tail -2 gpurun_out/r2f/assign_$f.txt; done
