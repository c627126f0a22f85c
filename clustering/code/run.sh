#!/usr/bin/env bash
# stage 5 of the pipeline (README step 5): features -> cluster assignments, on the GPU hot path
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
DATA="$HERE/../../data"
python "$HERE/cli.py" cluster --feature_path="$DATA/features/shard-000000.pkl" \
  --out_path="$DATA/clusters" --meta_path="$DATA/videos" "$@"
