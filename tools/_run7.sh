cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
for a in 0 1 2; do ACAV_FILTER_ABL=$a timeout 300 python tools/run_assign_only.py 1000000 20 filter 2>&1 | grep "filter kernel" | sed "s/^/abl $a: /"; done
