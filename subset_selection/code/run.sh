#!/usr/bin/env bash
# stage 6 of the pipeline: cluster assignments -> output.csv (greedy MI subset), on the GPU hot path
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
DATA="$HERE/../../data"
python "$HERE/cli.py" run --shards_path="$DATA/clusters/shard-000000.pkl" \
  --meta_path="$DATA/videos" --out_path="$DATA/output.csv" "$@"
