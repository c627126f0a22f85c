"""From a rocprofv3 kernel trace of the greedy loop: duration of the content stream's launches (k_fy_gather_select_multi) and the idle gap
between one launch's end and the next one's start on that stream."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_fy_gather_select_multi" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
s = np.array([int(r["Start_Timestamp"]) for r in rows], np.int64)
e = np.array([int(r["End_Timestamp"]) for r in rows], np.int64)
dur, gap, period = (e - s) / 1e3, (s[1:] - e[:-1]) / 1e3, (s[1:] - s[:-1]) / 1e3
lo = len(dur) // 4
print("launches", len(dur), "duration us: mean %.2f median %.2f" % (dur[lo:].mean(), np.median(dur[lo:])),
      "| gap to the next launch us: mean %.2f median %.2f p90 %.2f" % (gap[lo:].mean(), np.median(gap[lo:]), np.percentile(gap[lo:], 90)),
      "| start-to-start us: mean %.2f" % period[lo:].mean())
# every 16th launch follows a cross-stream event wait
idx = np.arange(len(gap))
print("gap by position in the group of 16:", " ".join("%.1f" % gap[lo:][(idx[lo:] % 16) == q].mean() for q in range(16)))

# the position stream: part -> tile -> resolve per group of 16 iterations
side = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("k_fy_part_multi", "k_fy_tile_multi", "k_fy_resolve_multi"))]
side.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["part" if "part" in r["Kernel_Name"] else "tile" if "tile" in r["Kernel_Name"] else "resolve" for r in side]
ss = np.array([int(r["Start_Timestamp"]) for r in side], np.int64)
se = np.array([int(r["End_Timestamp"]) for r in side], np.int64)
lo2 = len(side) // 4
for a, b in (("part", "tile"), ("tile", "resolve"), ("resolve", "part")):
    g = [(ss[i + 1] - se[i]) / 1e3 for i in range(lo2, len(side) - 1) if names[i] == a and names[i + 1] == b]
    print("position stream: gap %s -> %s us: mean %.1f median %.1f" % (a, b, np.mean(g), np.median(g)))
for a in ("part", "tile", "resolve"):
    dd = [(se[i] - ss[i]) / 1e3 for i in range(lo2, len(side)) if names[i] == a]
    print("position stream: %s duration us: mean %.1f" % (a, np.mean(dd)))
pp = [ss[i] for i in range(lo2, len(side)) if names[i] == "part"]
print("position stream: part-to-part period us: mean %.1f" % (np.diff(pp).mean() / 1e3))
