"""Near-tie census shared by the CPU (oracle) and GPU tests of the BASELINE-shape goldens `kmeans_big_*`:
where another fp32 evaluation of sgd_clustering.py:63-79 (the reference's GEMM order vs the canonical segment chain)
picks a different centre, the two centres must be closer in EXACT arithmetic than fp32 can resolve."""
import hashlib
import os
import sys

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)
import synth  # noqa: E402

U = 2.0 ** -24  # fp32 unit roundoff


def load_case(name):
    g = np.load(os.path.join(GOLDEN, f"kmeans_big_{name}.npz"))
    n, d, k = int(g["N"]), int(g["d"]), int(g["K"])
    x = synth.overlapping_rows(int(g["data_seed"]), n, d, int(g["comps"]), float(g["spread"]))
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["x_sha256"]), "the rows are not the generator's rows"
    return g, x, n, d, k


def bisector(g, x, centers):
    xb = synth.bisector_rows(x, centers, g["bis_idx"], g["bis_i"], g["bis_j"], g["bis_t"])
    assert hashlib.sha256(xb.tobytes()).hexdigest() == str(g["bis_sha256"]), "bisector rows differ from the generator's"
    return xb


def fp32_bound(x64, c64, d):
    """Any fp32 evaluation of fl(fl(-2 c.x + |x|^2) + |c|^2): the dot in ANY summation order is within
    gamma_d |c||x| (gamma_d ~ d u, Cauchy-Schwarz on sum |c_j x_j|), each squared norm (sum, sqrt, square) within
    (d + 4) u of itself, the two final additions within u of their results.  [rows] float64."""
    nx, nc = np.sqrt((x64 ** 2).sum(1)), np.sqrt((c64 ** 2).sum(1))
    return 2 * (d + 2) * U * nc * nx + (d + 4) * U * (nx ** 2 + nc ** 2) + 2 * U * (nx + nc) ** 2


def census(x, centers, counts, count, reinit, got, want, tag, max_ulps=None):
    """got / want: labels of two fp32 evaluations.  Every row where they differ must be an exact-arithmetic near-tie of
    those two centres: |D(got) - D(want)| <= bound(got) + bound(want), D in float64 incl. the under-use division.
    Prints the count (report it, do not hide it); returns (mismatches, max |gap| / bound)."""
    got, want = np.asarray(got).astype(np.int64), np.asarray(want).astype(np.int64)
    bad = np.nonzero(got != want)[0]
    worst = 0.0
    if len(bad):
        k, d = centers.shape
        thr = np.float32((count / k) ** reinit[0])
        div = np.where(counts < thr, float(reinit[1]), 1.0)
        x64 = x[bad].astype(np.float64)
        gaps, bounds = [], []
        for lab in (got[bad], want[bad]):
            c64 = centers[lab].astype(np.float64)
            dist = ((x64 - c64) ** 2).sum(1) / div[lab]
            gaps.append(dist)
            bounds.append(fp32_bound(x64, c64, d) / div[lab])
        gap, bound = np.abs(gaps[0] - gaps[1]), bounds[0] + bounds[1]
        worst = float((gap / bound).max())
        # the empirical statement is far tighter than the worst-case bound: in units of the fp32 spacing of the
        # compared distances themselves (no fp32 result can separate two values closer than that)
        ulps = gap / np.spacing(np.maximum(gaps[0], gaps[1]).astype(np.float32)).astype(np.float64)
        assert (gap <= bound).all(), (f"{tag}: {int((gap > bound).sum())} label differences are NOT near-ties "
                                      f"(largest exact gap {gap.max():.3e} vs fp32 bound {bound[gap.argmax()]:.3e})")
        print(f"[near-tie census] {tag}: {len(bad)} of {len(got)} labels differ from the reference, all at exact gaps "
              f"<= {gap.max():.3e} = {ulps.max():.2f} ulp of the distances (worst-case fp32 bound there "
              f"{bound[gap.argmax()]:.3e}, worst gap/bound {worst:.2e})")
        assert max_ulps is None or ulps.max() <= max_ulps, f"{tag}: a label difference at {ulps.max():.2f} ulp"
    else:
        print(f"[near-tie census] {tag}: 0 of {len(got)} labels differ from the reference")
    return len(bad), worst
