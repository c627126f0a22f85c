"""Dump one steady-state period of the greedy loop's kernel trace (all streams), times relative to a k_fy_resolve end."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
res = [i for i, r in enumerate(rows) if "k_fy_resolve_multi" in r["Kernel_Name"]]
i0 = res[len(res) // 2]
t0 = int(rows[i0]["End_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > t0 - 150_000 and s < t0 + 560_000:
        n = r["Kernel_Name"]; k = n.find("k_")
        print("%+8.1f .. %+8.1f  (%6.1f us)  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), n[k:k + 28]))
