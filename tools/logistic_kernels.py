#!/usr/bin/env python3
"""The kernels that evaluate the hashed logistic objective (BASELINE config 3), alone and back to back, against the launch grid:
the probe (2r), the plain evaluation (1r 1w), the line evaluation (2r 2w), the OWL-QN trial (3r 3w) and its first-trial form
(3r 4w).  Every timing runs in a fresh context, so LBFGS_HIP_GRID_X32_K4 (workgroups per 32 CUs of the evaluation class; 0 = the
operators' own) can be swept inside one process:   python tools/logistic_kernels.py [n] [x32 ...]"""
import os
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import rust_lbfgs_amd as R  # noqa: E402
from rust_lbfgs_amd import hotpath as H, objectives  # noqa: E402
from rust_lbfgs_amd.math import DeviceVec  # noqa: E402


def timed(ctx, fn, reps=200):
    for _ in range(20):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e6


n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
for x32 in [int(v) for v in (sys.argv[2:] or ["0"])]:
    if x32:
        os.environ["LBFGS_HIP_GRID_X32_K4"] = str(x32)
    else:
        os.environ.pop("LBFGS_HIP_GRID_X32_K4", None)
    with R.Context(n) as ctx:
        vs = [DeviceVec(ctx) for _ in range(7)]
        x, xp, d, g, pg, wp, sp = vs
        xp.fill(0.3), d.fill(-0.1), wp.fill(1.0), pg.fill(0.0)
        lg = objectives.Logistic()
        row = {
            "probe 2r": (16, timed(ctx, lambda: H.objective_line_probe(lg, xp, d, 0.5))),
            "eval 1r1w": (16, timed(ctx, lambda: H.objective_eval(lg, xp, g))),
            "line eval 2r2w": (32, timed(ctx, lambda: H.objective_line_eval(lg, x, xp, d, 0.5, g))),
            "owl trial 3r3w": (48, timed(ctx, lambda: H.objective_owlqn_line_eval(lg, x, xp, d, 0.5, wp, g, pg, 0.5, 0, n))),
            "owl first 3r4w": (56, timed(ctx, lambda: H.objective_owlqn_first_trial(lg, x, xp, d, 0.5, wp, g, pg, 0.5, 0, n))),
        }
        print(f"n = {n}, K4 grid x32 = {x32 or 'own'}: " + "; ".join(f"{k} {us:.1f} us = {b * n / us / 1e6:.2f} TB/s" for k, (b, us) in row.items()),
              flush=True)
        for v in vs:
            v.free()
