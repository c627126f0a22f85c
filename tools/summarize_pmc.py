"""Fold the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the assign sweep into <prefix>_pmc_assign<tag>.json:
HBM bytes of ONE sweep (acav_kmeans_assign over the whole partition), corrected as MI355X_MICROARCH.md (HBM section)
prescribes.

A sweep is several launches -- the filter over all rows, for K > 256 k_assign_merge, then the emission-pass
instantiation of the filter, k_assign_cand and k_assign_f32 over the (usually empty) lists of undecided rows -- and TWO of
them are instantiations of the same template (`k_assign_f16_rw<..., 0>` (k_assign_bf16_rw before round 5) over 1M rows, `<..., 1>` over a handful).  Round
4's version keyed the launches by the bare kernel name and averaged the two instantiations into one "mean per launch"
(traffic 0.53 x algorithmic: impossible).  Now: every launch is keyed by its FULL instantiation, the number of sweeps is
the launch count of the instantiation that moves the most bytes, and the traffic of a sweep is the sum over every launch
of the pass divided by that count.  The summary refuses to be written when the result is below 0.99 x the algorithmic
bytes (rows read once + labels written once)."""
import csv
import json
import re
import sys

out, prefix = sys.argv[1], sys.argv[2]
# optional: tag of the pass pair (files <prefix>_pmc_fetch<tag>_..., <prefix>_pmc_write<tag>_...) and the shape they ran
tag = sys.argv[3] if len(sys.argv) > 3 else ""
N, D, K = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1_000_000, 1024, 256)


def instantiation(name):
    """'void (anonymous namespace)::k_assign_bf16_rw<true, 8, true, 3, 2, 0>(float const*, ...)' -> 'k_assign_bf16_rw<true, 8, true, 3, 2, 0>'"""
    m = re.search(r"(k_[A-Za-z0-9_]+)(<[^>(]*>)?", name)
    return None if not m else m.group(1) + (m.group(2) or "")


def per_instantiation(path):
    """-> {instantiation: [counter value of every launch]}"""
    acc = {}
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError:
        return acc
    for r in rows:
        key = instantiation(r["Kernel_Name"])
        if key is not None and key.startswith("k_assign"):
            acc.setdefault(key, []).append(float(r["Counter_Value"]))
    return acc


fetch = per_instantiation(f"{out}/{prefix}_pmc_fetch{tag}_counter_collection.csv")
write = per_instantiation(f"{out}/{prefix}_pmc_write{tag}_counter_collection.csv")
if not fetch or not write:
    sys.exit(f"no k_assign launches in {out}/{prefix}_pmc_(fetch|write){tag}_counter_collection.csv")
main = max(fetch, key=lambda k: sum(fetch[k]) / len(fetch[k]))  # the filter over all rows
sweeps_f, sweeps_w = len(fetch[main]), len(write.get(main, ()))
assert sweeps_f > 0 and sweeps_w > 0, (main, sweeps_f, sweeps_w)
# FETCH_SIZE counts 128-byte requests at 64 B for wide coalesced reads on gfx950 -> x 2; both counters are in KB
fetch_kb = sum(sum(v) for v in fetch.values()) / sweeps_f
write_kb = sum(sum(v) for v in write.values()) / sweeps_w
traffic = (2.0 * fetch_kb + write_kb) * 1024.0
alg = N * D * 4 + N * 8
ratio = traffic / alg
per_launch = {k: {"launches_per_sweep": len(v) / sweeps_f, "FETCH_SIZE_KB_mean": sum(v) / len(v),
                  "WRITE_SIZE_KB_mean": (sum(write[k]) / len(write[k])) if k in write else None} for k, v in fetch.items()}
assert ratio >= 0.99, (f"traffic {traffic:.4g} B per sweep is below the algorithmic {alg:.4g} B (x {ratio:.3f}): the rows alone are "
                       f"1.0 x -- launches are being mis-attributed", per_launch)
json.dump({
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex k_assign -- python tools/run_assign_only.py "
               f"{N} 5 filter {D} {K} (separate passes; tools/collect_profiles.sh)",
    "kernel": "k_assign_f16_rw (+ k_assign_merge for K > 256) + the launches over the undecided rows (emission-pass "
              "instantiation, k_assign_cand, k_assign_f32: empty lists on this data)",
    "rows": N, "d": D, "K": K, "sweeps_in_the_pass": sweeps_f, "main_instantiation": main,
    "per_instantiation": per_launch,
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads "
                  "on gfx950 -> x2; units are KB; SUM over every launch of one sweep (keyed by full template instantiation)",
    "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": ratio,
}, open(f"{out}/{prefix}_pmc_assign{tag}.json", "w"), indent=1)
print(open(f"{out}/{prefix}_pmc_assign{tag}.json").read())
