#!/usr/bin/env python3
"""Is a seed that fails the random sweep a defect or a run that amplifies last-bit differences?  For each named seed: row by
row, the oracle's own scatter under four other summation orders and from two last-bit neighbours of x0 (tests/fuzz_common.py order_sensitivity), the deviation of the
HIP path from the oracle under TWO partitions of the vectors among workgroups (the default grid with the persistent two-loop
kernel; one workgroup and a kernel per step: two more summation orders, both on the GPU -- the two launch forms of the
two-loop on the SAME grid agree bit for bit at these sizes and would not be a second sample), and the deviation of those two
from EACH OTHER.  A defect shows as a deviation from the oracle that the two GPU runs share; chaos shows as the two GPU runs
parting from each other as far as from the oracle.
    python tools/fuzz_outlier_diag.py 28675 57836 79105"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from rust_lbfgs_amd import objectives  # noqa: E402
from tests import fuzz_common as F  # noqa: E402


def dev(a, b, ro):
    f0 = max(abs(ro[0][3]), 1e-3)
    g0 = max(ro[0][5], 1e-6)
    return max((0.0 if (u != u and v != v) else abs(u - v) / s) for u, v, s in zip(a[3:], b[3:], F.scales(a, f0, g0)))


verdicts = []
for seed in [int(s) for s in sys.argv[1:]]:
    c = F.make_case(seed)
    ro, xo, eo = F.run_oracle(c, 0)
    floors, stable, all_stable = F.order_sensitivity(c, ro, eo)
    got = {}
    for vf in (False, True):
        for form, res, grid in (("resident", "1", None), ("per_step", "0", "1")):
            os.environ["LBFGS_HIP_RESIDENT"] = res
            os.environ.pop("LBFGS_HIP_GRID", None)
            if grid:
                os.environ["LBFGS_HIP_GRID"] = grid
            c["vector_free"] = vf
            got[(vf, form)] = F.run_product(R, objectives, c)
    c["vector_free"] = False
    print(f"seed {seed}: {c}\n  oracle: {len(ro)} rows, error {eo}; stable prefix {stable}, all stable {all_stable}")
    worst = None
    for i in range(len(floors)):
        a = ro[i]
        cells = []
        for vf in (False, True):
            r1, r2 = got[(vf, "resident")][0], got[(vf, "per_step")][0]
            if i >= len(r1) or i >= len(r2):
                cells.append(" (stopped)")
                continue
            dec = "" if tuple(r1[i][:3]) == tuple(a[:3]) == tuple(r2[i][:3]) else f" decisions {r1[i][:3]} {r2[i][:3]}"
            d1, d2, d12 = dev(a, r1[i], ro), dev(a, r2[i], ro), dev(r1[i], r2[i], ro)
            tol = max(1e-10, 20.0 * floors[i]) * (50.0 if vf else 1.0)
            flag = " <-- over" if max(d1, d2) > tol or dec else ""
            if flag and worst is None:
                worst = (i, vf, d1, d2, d12, bool(dec))
            cells.append(f" | {'vector-free' if vf else 'two-loop'}: default grid {d1:.1e} one workgroup {d2:.1e} between them {d12:.1e}{dec}{flag}")
        print(f"  row {i} {a[:3]} fx {a[3]:.6e} floor {floors[i]:.1e}" + "".join(cells))
    if worst:
        i, vf, d1, d2, d12, dec = worst
        kind = ("the two GPU runs part from each other as far as from the oracle: amplified summation order"
                if (dec or d12 >= 0.05 * max(d1, d2)) else "BOTH GPU runs deviate together: look for a defect")
        verdicts.append((seed, i, "vector-free" if vf else "two-loop", kind))
        print(f"  => first row over its tolerance: {i} ({'vector-free' if vf else 'two-loop'}): {kind}")
    else:
        verdicts.append((seed, None, "", "within tolerance in both runs"))
print()
for v in verdicts:
    print("seed %d: row %s %s: %s" % v)
