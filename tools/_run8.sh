cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_cli.py tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err
tail -4 gpurun_out/r2h/pytest.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2h/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step']); print(d['stages']); print(d['roofline']['frac'], d['roofline']['sweep_ms'])
PY
tail -3 gpurun_out/r2h/bench.err
