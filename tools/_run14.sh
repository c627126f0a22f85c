cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 1000000 256 2 0 20000 2>&1 | grep "acav" | cut -c1-220
ACAV_MI_TIMING=1 timeout 300 python tools/bench_mi.py 100000 256 2 2>&1 | grep "acav" | cut -c1-220
timeout 300 python tools/bench_mi_lockstep.py 100000 256 2 10 2>&1 | tail -1 | cut -c1-330
