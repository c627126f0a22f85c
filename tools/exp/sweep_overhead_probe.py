"""Where do the ~100 us between the sweep's events and the filter kernel's own time go?  Warm handle, repeated sweeps, the
library's event timer around the whole acav_kmeans_assign call vs the filter kernel's own events.  Run with GPU_MAX_HW_QUEUES=4 / 16."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd import _lib
from acav100m_amd.clustering import KMeans
lib = acav100m_amd.load_library()
n, d, k = 1_000_000, 1024, 256
gen = torch.Generator(device="cuda").manual_seed(0)
cen = torch.randn(k, d, device="cuda", generator=gen)
x = cen[torch.randint(0, k, (n,), device="cuda", generator=gen)] + 0.3 * torch.randn(n, d, device="cuda", generator=gen)
lab = torch.empty(n, dtype=torch.long, device="cuda")
torch.cuda.synchronize()
for fresh in (False, True):
    km = None
    out = []
    for rep in range(6):
        if km is None or fresh:
            km = KMeans(None, d, k)
            km.centers, km.counts, km.count = cen.cpu().numpy(), np.full(k, 1000, np.float32), 10 * k + n
            km.to("cuda:0")
            km.synchronize()
        _lib.check(lib.acav_kmeans_timer_begin(km._h))
        _lib.check(lib.acav_kmeans_assign(km._h, _lib.ptr(x), n, _lib.ptr(lab), None))
        ms = C.c_float(0)
        _lib.check(lib.acav_kmeans_timer_end(km._h, C.byref(ms)))
        fm = C.c_float(0)
        _lib.check(lib.acav_kmeans_filter_time(km._h, C.byref(fm)))
        out.append("%.0f/%.0f" % (ms.value * 1e3, fm.value * 1e3))
    print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "fresh handle per sweep" if fresh else "one warm handle", "sweep/filter us:", " ".join(out), flush=True)

# back-to-back sweeps get slower (789 -> 1176 us over six sweeps), sweeps with host work in between do not: clocks?
import time
km = KMeans(None, d, k)
km.centers, km.counts, km.count = cen.cpu().numpy(), np.full(k, 1000, np.float32), 10 * k + n
km.to("cuda:0")
km.synchronize()
for pause in (0.0, 0.002, 0.02):
    out = []
    for rep in range(14):
        _lib.check(lib.acav_kmeans_assign(km._h, _lib.ptr(x), n, _lib.ptr(lab), None))
        km.synchronize()
        fm = C.c_float(0)
        _lib.check(lib.acav_kmeans_filter_time(km._h, C.byref(fm)))
        out.append("%.0f" % (fm.value * 1e3))
        if pause:
            time.sleep(pause)
    print("pause %.0f ms between sweeps, filter us:" % (pause * 1e3), " ".join(out), flush=True)
