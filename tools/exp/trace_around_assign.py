"""Print the kernels of a rocprofv3 kernel trace that run within 400 us before / after each filter launch (what is inside the sweep's events?)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_assign_f16_rw" in r["Kernel_Name"]]
for i in idx:
    t0, t1 = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print("---- filter launch: %.1f us" % ((t1 - t0) / 1e3))
    for r in rows[max(0, i - 12): i + 8]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > t0 - 400_000 and e < t1 + 400_000:
            name = r["Kernel_Name"]
            k = name.find("k_")
            print("   %+9.1f .. %+9.1f us  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, r.get("Queue_Id", "?"), name[k if k >= 0 else 0:][:60]))
