#!/usr/bin/env bash
# Curation hot path only (the stages upstream of the features are out of scope).
# Unlike the reference's top-level run.sh this also runs the clustering stage, which the reference
# script omits although subset selection needs its output (README.md:118-124).
set -e
bash ./clustering/code/run.sh
bash ./subset_selection/code/run.sh
