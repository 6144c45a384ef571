import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import rust_lbfgs_amd as R
from rust_lbfgs_amd.math import DeviceVec
ctx = R.Context(1 << 20)
x = DeviceVec(ctx); x.fill(1.0)
for _ in range(100): ctx.scalars(0, 2)
t0 = time.perf_counter()
N = 2000
for _ in range(N): ctx.scalars(0, 2)
t1 = time.perf_counter()
print("scalars_read on an idle stream: %.1f us" % ((t1 - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): x.vecdot_slot(x, 3); ctx.scalars(3, 1)
t1 = time.perf_counter()
print("tiny dot kernel + scalars_read:  %.1f us" % ((t1 - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): x.vecdot_slot(x, 3)
ctx.sync()
t1 = time.perf_counter()
print("tiny dot kernel launch only (async, amortised): %.1f us" % ((t1 - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): ctx.sync()
t1 = time.perf_counter()
print("stream sync on an idle stream: %.1f us" % ((t1 - t0) / N * 1e6))
