cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_configs.py -m gpu -x -q -k "not cfg3 and not cfg1" > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log
for v in 0 1; do ACAV_FILTER_V1=$v timeout 300 python tools/run_assign_only.py 1000000 20 filter > gpurun_out/r2f/assign_v1_$v.txt 2>&1; done
tail -4 gpurun_out/r2f/pytest.log; tail -3 gpurun_out/r2f/assign_v1_0.txt; tail -3 gpurun_out/r2f/assign_v1_1.txt
