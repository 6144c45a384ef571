#!/usr/bin/env python3
"""Is the trial probe's distance from the plain two-read kernels the objective's arithmetic?  Same skeleton, same bytes (two
n-vectors read, two sums), three element operators: OpNorms2 (two multiply-adds), the probe of the hashed quadratic (two
splitmix64 per element, a few flops) and of the hashed logistic (the same hashes + exp, log1p, a division).   python tools/probe_alu_check.py [n ...]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import rust_lbfgs_amd as R  # noqa: E402
from rust_lbfgs_amd import hotpath as H, objectives  # noqa: E402
from rust_lbfgs_amd.math import DeviceVec  # noqa: E402


def timed(ctx, fn, reps=300):
    for _ in range(20):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e6


for n in [int(float(v)) for v in (sys.argv[1:] or ["12500224", "10000000", "100000000"])]:
    with R.Context(n) as ctx:
        x, d = DeviceVec(ctx), DeviceVec(ctx)
        x.fill(0.3)
        d.fill(-0.1)
        row = {"norms (OpNorms2)": timed(ctx, lambda: H.norms_sq(x, d)),
               "probe, hashed quadratic": timed(ctx, lambda: H.objective_line_probe(objectives.Quadratic(), x, d, 0.5)),
               "probe, hashed logistic": timed(ctx, lambda: H.objective_line_probe(objectives.Logistic(), x, d, 0.5))}
        print(f"n = {n}: " + "; ".join(f"{k} {v:.1f} us = {16.0 * n / v / 1e6:.2f} TB/s" for k, v in row.items()), flush=True)
        x.free()
        d.free()
