#!/usr/bin/env python3
"""Hunt for a non-deterministic result of the resident two-loop kernel: the same recursion many times over, each time in a
fresh context, against the kernel-per-step path on the same device data.   python tools/resident_repro.py n m k end repeats"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from rust_lbfgs_amd import hotpath as H, objectives  # noqa: E402
from rust_lbfgs_amd.math import DeviceVec  # noqa: E402

n, m, k, end, reps = (int(v) for v in sys.argv[1:6])


def run(resident):
    os.environ["LBFGS_HIP_RESIDENT"] = "1" if resident else "0"
    with R.Context(n) as ctx:
        hist = H.History(ctx, m)
        g, d, tmp = (DeviceVec(ctx) for _ in range(3))
        q = objectives.Quadratic()
        for j in range(m):
            tmp.fill(0.25 + 0.1 * j)
            H.objective_eval(q, tmp, hist.s(j), 0)
            H.objective_eval(q, hist.s(j), hist.y(j), 0)
            hist.y(j).vecadd(hist.s(j), 2.0)
        ys = [hist.y(j).vecdot(hist.s(j)) for j in range(m)]
        hist.set_scalars(ys=np.array(ys), alpha=np.zeros(m))
        tmp.fill(-0.3)
        H.objective_eval(objectives.Logistic(), tmp, g, 0)
        ctx.set_scalars(7, [ys[end], hist.y(end).vecdot(hist.y(end))])
        outs = []
        for _ in range(3):
            hist.set_scalars(alpha=np.zeros(m))
            hist.two_loop(d, g, k, end, 7, 8, 12)
            outs.append((d.to_numpy(), ctx.scalars(12, 2).copy(), hist.scalars()[1].copy()))
        res = ctx.resident_two_loops()
        hist.free()
        for v in (g, d, tmp):
            v.free()
    return outs, res


ref, _ = run(False)
bad = 0
for it in range(reps):
    outs, res = run(True)
    for j, (dv, dn, al) in enumerate(outs):
        err = np.max(np.abs(dv - ref[0][0])) / np.max(np.abs(ref[0][0]))
        if err > 1e-12:
            bad += 1
            diff = np.nonzero(np.abs(dv - ref[0][0]) > 1e-12 * np.max(np.abs(ref[0][0])))[0]
            print(f"rep {it} launch {j}: resident launches {res}, rel err {err:.3e}; {len(diff)} of {n} elements differ, first {diff[0]} last {diff[-1]}; "
                  f"alpha rel diff {np.abs(al - ref[0][2]) / np.maximum(np.abs(ref[0][2]), 1e-300)}; dn {dn} vs {ref[0][1]}", flush=True)
print(f"n={n} m={m} k={k} end={end}: {bad} bad launches of {3 * reps}")
