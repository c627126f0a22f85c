cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_cli.py tests/test_gpu_configs.py -m gpu -x -q -k "not cfg2 and not k1024" > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
ACAV_FY_ECAP=64 timeout 900 python -m pytest tests/test_gpu_mi.py -m gpu -x -q > gpurun_out/r2c/pytest_ecap.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest_ecap.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2c/ktmi -o kt -- python tools/bench_mi.py 1000000 256 2 0 3000 > gpurun_out/r2c/mi_1m.json 2>/dev/null
cp gpurun_out/r2c/ktmi/*kernel_stats.csv gpurun_out/r2c/mi_1m_kernel_stats.csv; rm -rf gpurun_out/r2c/ktmi
timeout 300 python tools/bench_mi.py 100000 256 2 > gpurun_out/r2c/mi_100k.json 2>&1
ACAV_FY_LEGACY=1 timeout 300 python tools/bench_mi.py 100000 256 2 > gpurun_out/r2c/mi_100k_legacy.json 2>&1
tail -5 gpurun_out/r2c/pytest.log; tail -3 gpurun_out/r2c/pytest_ecap.log; tail -1 gpurun_out/r2c/mi_1m.json; tail -1 gpurun_out/r2c/mi_100k.json; tail -1 gpurun_out/r2c/mi_100k_legacy.json
python - <<'PY'
import csv
for r in csv.reader(open('gpurun_out/r2c/mi_1m_kernel_stats.csv')):
    print(r[0][:50].ljust(50), r[1:5])
PY
