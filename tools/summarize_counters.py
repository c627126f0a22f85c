"""Fold the per-counter rocprofv3 --pmc passes of the assign filter (tools/collect_profiles.sh) into one JSON:
mean per launch of every counter, for the row-wave kernel (rw) and the round-1 layout (v1)."""
import csv
import glob
import json
import os
import re
import sys

out, prefix = sys.argv[1], sys.argv[2]
res = {"kernel": "k_assign_f16_rw (wave = 32 rows x 256 centres; k_assign_bf16_rw before round 5)",
       "workload": "tools/run_assign_only.py 1000000 5 filter (1M x 1024, K = 256)", "mean_per_launch": {}}
for path in sorted(glob.glob(os.path.join(out, f"{prefix}_pmc_*_counter_collection.csv"))):
    m = re.search(rf"{prefix}_pmc_(rw|v1)_([A-Z_]+)_counter_collection", os.path.basename(path))
    if not m:
        continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "k_assign_f16" in r["Kernel_Name"] or "k_assign_bf16" in r["Kernel_Name"]]
    if vals:
        res["mean_per_launch"].setdefault(m.group(1), {})[m.group(2)] = sum(vals) / len(vals)
print(json.dumps(res, indent=1))
