cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
ACAV_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2n/bench2.json 2> gpurun_out/r2n/bench2.err
echo "rc=$?"; tail -c 1500 gpurun_out/r2n/bench2.json; tail -5 gpurun_out/r2n/bench2.err
