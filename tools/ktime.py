import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import rust_lbfgs_amd as R
from rust_lbfgs_amd.math import DeviceVec
n = int(sys.argv[1])
with R.Context(n) as ctx:
    u, v = DeviceVec(ctx), DeviceVec(ctx)
    u.fill(1.0); v.fill(2.0)
    for _ in range(50): u.vecdot_slot(v, 100)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(2000): u.vecdot_slot(v, 100)
    ctx.sync()
    print(sys.argv[2], "dot kernel spacing us:", (time.perf_counter() - t0) / 2000 * 1e6)
