#!/usr/bin/env python3
"""Small-vector regime (config 5: n = 3e6, 24 MB vectors -- they fit the Infinity Cache whole): A/B of the launch grid for the
kernels that surround the two-loop there -- OpDot, OpLineStep, OpHistUpdate<damping>, OpNorms2.  Per grid: microseconds per
launch from HIP events around every launch (ctx.prof) and from the wall clock over a back-to-back batch.
    python tools/small_vector_ab.py [n ...]   (default 3000000 1000000)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from rust_lbfgs_amd import _ffi, hotpath as H  # noqa: E402
from rust_lbfgs_amd.math import DeviceVec  # noqa: E402

REPS = 400
for n in [int(v) for v in sys.argv[1:]] or [3_000_000, 1_000_000]:
    print(f"## n = {n} ({8 * n / 1e6:.0f} MB vectors), {REPS} launches per figure: us per launch by HIP events / by wall clock")
    with R.Context(n) as ctx:
        x, xp, g, gp, d = (DeviceVec(ctx, np.random.default_rng(k).standard_normal(n)) for k in range(5))
        hist = H.History(ctx, 1)
        ops = {
            "OpDot (2r)": (lambda: g.vecdot_slot(d, 40), _ffi.K_BLAS1, 16.0),
            "OpLineStep (2r 1w)": (lambda: H.line_step(x, xp, d, 0.25), _ffi.K_LINE, 24.0),
            "OpHistUpdate<damping> (4r 2w)": (lambda: hist.update(0, x, xp, g, gp, 0.5, True, 50), _ffi.K_UPDATE, 48.0),
            "OpNorms2 (2r)": (lambda: H.norms_sq(x, g, 60), _ffi.K_BLAS1, 16.0),
        }
        for name, (fn, kclass, bpe) in ops.items():
            row = []
            for grid in (0, 108, 216, 432, 864, 1728):
                ctx.set_grid(grid)
                for _ in range(20):
                    fn()
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(REPS):
                    fn()
                ctx.sync()
                wall = (time.perf_counter() - t0) / REPS * 1e6
                ctx.prof_enable(True); ctx.prof_reset()
                for _ in range(REPS):
                    fn()
                cnt, ms = ctx.prof_read(kclass)
                ctx.prof_enable(False)
                ev = ms / max(cnt, 1) * 1e3
                row.append(f"{'default' if grid == 0 else grid}: {ev:5.1f} / {wall:5.1f}")
            best = bpe * n / 1e6
            print(f"{name:32s} ({best / 8.0:5.1f} us at 8 TB/s) " + " | ".join(row))
        ctx.set_grid(0)
        hist.free()
        for v in (x, xp, g, gp, d):
            v.free()
