"""Why the K <= 256 assign filter is slower back to back than after a pause.

Runs tools/exp/sustained_bench (the product's filter tile launched N times with nothing between the launches, a one-wave shader
clock probe beside them) in several regimes while THIS process samples the SMU's view of the device (amdsmi gpu_metrics:
gfx clock per XCD, memory clock, socket power, hotspot / memory temperature, throttle status and the PPT / thermal residency
accumulators; sysfs hwmon as the fallback), then the product library's own sweep through the C ABI.

  python tools/filter_sustained.py [--launches 400] [--out gpurun_out/r06_filter_sustained]

Writes <out>.json (everything) and <out>.txt (the table DESIGN quotes).  Experiment tooling: never imported by the product."""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Sampler(threading.Thread):
    """~5 ms samples of the device's clocks / power / temperatures: amdsmi when it works, hwmon sysfs files otherwise."""

    def __init__(self, period=0.005):
        super().__init__(daemon=True)
        self.period, self.rows, self.stop_flag = period, [], False
        self.kind, self.h, self.files, self.static = None, None, [], {}
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h, self.smi, self.kind = hs[0], amdsmi, "amdsmi"
            m = amdsmi.amdsmi_get_gpu_metrics_info(self.h)
            self.static["metrics_keys"] = sorted(m.keys())
            try:
                self.static["power_cap"] = amdsmi.amdsmi_get_power_cap_info(self.h)
            except Exception as exc:
                self.static["power_cap"] = repr(exc)
            self.static["handles"] = len(hs)
        except Exception as exc:
            self.static["amdsmi_error"] = repr(exc)
            pats = ("freq*_input", "power*_average", "power*_input", "temp*_input")
            for pat in pats:
                self.files += sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/" + pat))
            self.kind = "sysfs" if self.files else None

    KEYS = ("current_gfxclks", "current_gfxclk", "average_gfxclk_frequency", "current_uclk", "average_uclk_frequency",
            "current_socket_power", "average_socket_power", "temperature_hotspot", "temperature_mem", "temperature_vrsoc",
            "indep_throttle_status", "throttle_status", "average_gfx_activity", "average_umc_activity", "accumulation_counter",
            "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "prochot_residency_acc",
            "gfxclk_lock_status", "energy_accumulator", "firmware_timestamp")

    def sample(self):
        row = {"t_ns": time.time_ns()}
        if self.kind == "amdsmi":
            try:
                m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
                for k in self.KEYS:
                    if k in m:
                        v = m[k]
                        row[k] = [x for x in v if isinstance(x, (int, float))][:8] if isinstance(v, (list, tuple)) else v
            except Exception as exc:
                row["error"] = repr(exc)
        elif self.kind == "sysfs":
            for f in self.files:
                try:
                    row[f.split("/device/")[0].split("/")[-1] + ":" + os.path.basename(f)] = int(open(f).read().split()[0])
                except Exception:
                    pass
        row["dt_us"] = (time.time_ns() - row["t_ns"]) // 1000
        return row

    def run(self):
        while not self.stop_flag:
            self.rows.append(self.sample())
            time.sleep(self.period)

    def window(self, t0_ns, t1_ns):
        return [r for r in self.rows if t0_ns <= r["t_ns"] <= t1_ns]


def _num(v):
    if isinstance(v, (list, tuple)):
        v = [x for x in v if isinstance(x, (int, float)) and 0 < x < 65535]
        return sum(v) / len(v) if v else None
    return v if isinstance(v, (int, float)) and v not in (65535, 0xFFFFFFFF) else None


def summarise(rows, key):
    vals = [_num(r.get(key)) for r in rows]
    vals = [v for v in vals if v is not None]
    if not vals:
        return None
    return {"first": vals[0], "last": vals[-1], "min": min(vals), "max": max(vals), "mean": sum(vals) / len(vals), "n": len(vals)}


def residency(rows, key):
    """PVIOL / TVIOL as amdsmi.h defines them: delta of the residency accumulator over delta of the accumulation counter."""
    rs = [r for r in rows if isinstance(r.get(key), int) and isinstance(r.get("accumulation_counter"), int)]
    if len(rs) < 2:
        return None
    da = rs[-1]["accumulation_counter"] - rs[0]["accumulation_counter"]
    return None if da <= 0 else 100.0 * (rs[-1][key] - rs[0][key]) / da


def run_harness(sampler, exe, argv, label):
    t0 = time.time_ns()
    out = subprocess.run([exe] + [str(a) for a in argv], capture_output=True, text=True, timeout=300)
    t1 = time.time_ns()
    if out.returncode != 0:
        return {"label": label, "error": out.stdout[-400:] + out.stderr[-400:]}
    res = json.loads(out.stdout)
    res["label"] = label
    win = sampler.window(res["host_start_ns"], res["host_end_ns"])
    idle = sampler.window(t0, res["host_start_ns"])[-40:]
    res["smu"] = {k: summarise(win, k) for k in Sampler.KEYS if summarise(win, k) is not None}
    res["smu_idle_before"] = {k: summarise(idle, k) for k in ("current_gfxclks", "current_socket_power", "current_uclk", "temperature_hotspot")
                              if summarise(idle, k) is not None}
    res["pviol_pct"] = residency(win, "ppt_residency_acc")
    res["tviol_pct"] = residency(win, "socket_thm_residency_acc")
    res["hbm_tviol_pct"] = residency(win, "hbm_thm_residency_acc")
    res["smu_trace"] = [{"ms": (r["t_ns"] - res["host_start_ns"]) * 1e-6, "gfx_mhz": _num(r.get("current_gfxclks")),
                         "uclk_mhz": _num(r.get("current_uclk")), "power_w": _num(r.get("current_socket_power")),
                         "hot_c": _num(r.get("temperature_hotspot")), "throttle": r.get("indep_throttle_status")} for r in win[::4]]
    res["process_wall_s"] = (t1 - t0) * 1e-9
    return res


def product_sweeps(sampler, n, d, k, reps):
    """The product path itself: `reps` acav_kmeans_assign sweeps (filter + its zero-length re-check tail) on one handle with a
    synchronisation per sweep (bench.py's back_to_back figure), the filter launch's own event time per sweep."""
    import ctypes as C
    import numpy as np
    import torch
    import acav100m_amd
    from acav100m_amd import _lib
    from acav100m_amd.clustering import KMeans
    lib = acav100m_amd.load_library()
    gen = torch.Generator(device="cuda").manual_seed(0)
    cen = torch.randn(k, d, device="cuda", generator=gen)
    x = cen[torch.randint(0, k, (n,), device="cuda", generator=gen)] + 0.3 * torch.randn(n, d, device="cuda", generator=gen)
    lab = torch.empty(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    km = KMeans(None, d, k)
    km.centers, km.counts, km.count = cen.cpu().numpy(), np.full(k, 1000, np.float32), 10 * k + n
    km.to("cuda:0")
    _lib.check(lib.acav_kmeans_assign(km._h, _lib.ptr(x), n, _lib.ptr(lab), None))  # builds the centre copy
    km.synchronize()
    time.sleep(0.5)
    ms, t0 = [], time.time_ns()
    for _ in range(reps):
        _lib.check(lib.acav_kmeans_assign(km._h, _lib.ptr(x), n, _lib.ptr(lab), None))
        km.synchronize()
        fm = C.c_float(0)
        _lib.check(lib.acav_kmeans_filter_time(km._h, C.byref(fm)))
        ms.append(fm.value)
    t1 = time.time_ns()
    win = sampler.window(t0, t1)
    tail = sorted(ms[reps // 2:])
    bytes_ = n * d * 4 + n * 8
    return {"label": "product library, acav_kmeans_assign x %d (sync per sweep)" % reps, "launch_ms": [round(v, 4) for v in ms],
            "first_ms": ms[0], "settled_ms": tail[len(tail) // 2], "frac_first": bytes_ / ms[0] / 1e-3 / 8e12,
            "frac_settled": bytes_ / tail[len(tail) // 2] / 1e-3 / 8e12, "wall_ms_per_sweep": (t1 - t0) * 1e-6 / reps,
            "smu": {kk: summarise(win, kk) for kk in Sampler.KEYS if summarise(win, kk) is not None},
            "pviol_pct": residency(win, "ppt_residency_acc"), "tviol_pct": residency(win, "socket_thm_residency_acc"),
            "filter_stats": km.filter_stats()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_filter_sustained"))
    ap.add_argument("--skip-product", action="store_true")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    sampler = Sampler()
    sampler.start()
    time.sleep(0.3)
    exe, exe_nm = os.path.join(ROOT, "tools/exp/sustained_bench"), os.path.join(ROOT, "tools/exp/sustained_bench_nomfma")
    exe_nw8 = os.path.join(ROOT, "tools/exp/sustained_bench_nw8")
    exe_k1024, exe_k1024_nm = os.path.join(ROOT, "tools/exp/sustained_bench_k1024"), os.path.join(ROOT, "tools/exp/sustained_bench_k1024_nomfma")
    n, L = args.rows, args.launches
    runs = []
    plan = [(exe, [n, 1024, 256, L, 0, "mix"], "filter, back to back, mixture rows"),
            (exe, [n, 1024, 256, L // 4, 2000, "mix"], "filter, 2 ms pause between launches"),
            (exe_nm, [n, 1024, 256, L, 0, "mix"], "filter WITHOUT the MFMAs (timing-only ablation), back to back"),
            (exe, [n, 1024, 256, L, 0, "uni"], "filter, back to back, uniform rows"),
            (exe, [n, 1024, 256, L, 0, "mix"], "filter, back to back, mixture rows (repeat)"),
            (exe_nw8, [n, 1024, 256, L, 0, "mix"], "256-row tile (8 waves, centre ring 3, spread issue), back to back"),
            (exe_nw8, [n, 1024, 256, L // 4, 2000, "mix"], "256-row tile, 2 ms pause between launches"),
            # K = 1024 (the MFMA-bound regime of cfg4 / cfg5): is the (tile, group) pair kernel's clock capped too?
            (exe_k1024, [n, 1024, 1024, L // 2, 0, "mix"], "K = 1024 pair kernel, back to back"),
            (exe_k1024, [n, 1024, 1024, L // 8, 10000, "mix"], "K = 1024 pair kernel, 10 ms pause between launches"),
            (exe_k1024_nm, [n, 1024, 1024, L // 2, 0, "mix"], "K = 1024 pair kernel WITHOUT the MFMAs (timing-only ablation), back to back")]
    for e, a, label in plan:
        if not os.path.exists(e):
            runs.append({"label": label, "error": "not built: " + e})
            continue
        time.sleep(1.0)  # the device at rest between regimes
        runs.append(run_harness(sampler, e, a, label))
        r = runs[-1]
        print(label, "->", {k: r.get(k) for k in ("first_ms", "settled_ms", "frac_first", "frac_settled", "pviol_pct")}, flush=True)
    if not args.skip_product:
        try:
            time.sleep(1.0)
            runs.append(product_sweeps(sampler, n, 1024, 256, min(L, 200)))
        except Exception as exc:
            runs.append({"label": "product library", "error": repr(exc)})
    sampler.stop_flag = True
    sampler.join(timeout=2)
    doc = {"sampler": sampler.kind, "sampler_static": {k: (v if not isinstance(v, dict) else {a: str(b) for a, b in v.items()})
                                                        for k, v in sampler.static.items()},
           "samples": len(sampler.rows), "sample_cost_us_mean": sum(r["dt_us"] for r in sampler.rows) / max(1, len(sampler.rows)),
           "runs": runs}
    json.dump(doc, open(args.out + ".json", "w"), indent=1, default=str)
    with open(args.out + ".txt", "w") as f:
        f.write("K = 256 assign filter under sustained load (1M x 1024 fp32 rows, 4.104 GB algorithmic per launch); sampler: %s\n" % sampler.kind)
        f.write("%-76s %8s %8s %6s %6s %7s | %9s %9s %8s %8s %7s %7s\n" % ("regime", "first ms", "settled", "frac1", "fracS", "TFLOP/s", "probe GHz0", "probe GHzS",
                                                                           "gfx MHz", "power W", "PVIOL %", "hot C"))
        for r in runs:
            if "error" in r:
                f.write("%-76s ERROR %s\n" % (r["label"], r["error"][:200]))
                continue
            pg = r.get("probe_ghz") or []
            pm = r.get("probe_ms") or []
            # probe clock in the 300 ms idle lead-in vs the last third of the loaded window
            idle = [g for g, t in zip(pg, pm) if t < 250]
            load_end = 300 + (r.get("launch_at_ms") or [0])[-1]
            load = [g for g, t in zip(pg, pm) if 300 + 0.66 * (load_end - 300) < t < load_end]
            sm = r.get("smu", {})
            f.write("%-76s %8.4f %8.4f %6.3f %6.3f %7s | %9s %9s %8s %8s %7s %7s\n" % (
                r["label"][:76], r["first_ms"], r["settled_ms"], r["frac_first"], r["frac_settled"], "%.0f" % r["tflops_settled"] if "tflops_settled" in r else "-",
                "%.3f" % (sum(idle) / len(idle)) if idle else "-", "%.3f" % (sum(load) / len(load)) if load else "-",
                "%.0f" % sm["current_gfxclks"]["mean"] if "current_gfxclks" in sm else "-",
                "%.0f" % sm["current_socket_power"]["mean"] if "current_socket_power" in sm else "-",
                "%.1f" % r["pviol_pct"] if r.get("pviol_pct") is not None else "-",
                "%.0f" % sm["temperature_hotspot"]["max"] if "temperature_hotspot" in sm else "-"))
        f.write("\nper-launch ms, back to back, mixture rows (every 10th): ")
        r0 = runs[0]
        if "launch_ms" in r0:
            f.write(" ".join("%.3f" % v for v in r0["launch_ms"][::10]) + "\n")
            f.write("probe GHz at 1 ms means (every 10th): " + " ".join("%.2f" % v for v in (r0.get("probe_ghz") or [])[::10]) + "\n")
            f.write("SMU trace (ms from first launch: gfx MHz / uclk MHz / W / hotspot C): " +
                    " | ".join("%.0f: %s/%s/%s/%s" % (s["ms"], s["gfx_mhz"] and round(s["gfx_mhz"]), s["uclk_mhz"], s["power_w"], s["hot_c"])
                               for s in r0.get("smu_trace", [])[::3]) + "\n")
    print(open(args.out + ".txt").read())


if __name__ == "__main__":
    main()
