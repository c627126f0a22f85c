cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e/pytest.log
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 100000 256 2 > gpurun_out/r2e/mi_100k.log 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err
tail -4 gpurun_out/r2e/pytest.log; grep -h "acav\|us_per_iter" gpurun_out/r2e/mi_100k.log | cut -c1-250; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2e/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step']); print(d['stages']); print(d['roofline_mi']['achieved'], d['variants'], d['cpu_baseline']['value'])
PY
