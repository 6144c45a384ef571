#!/usr/bin/env python3
"""Does a vector that was just read come back faster than one from HBM?  (MI355X: 256 MiB Infinity Cache, memory side.)
    python tools/mall_probe.py [n]
Times `vecdot_slot(v, w)` (2 reads, nothing written, results stay on the device) with HIP-side wall time over 40 launches:
  cold    the pair cycles over 8 different vector pairs (1.6 GB at n = 12 500 224: nothing survives in any cache)
  hot     the same pair every time (200 MB)
  hot1    v.v of the same vector every time (100 MB)
under the default load policy of this size and with the `nt` hint forced on (LBFGS_HIP_NT_THRESHOLD_MB=1, second process)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n):
    import numpy as np  # noqa: F401

    import rust_lbfgs_amd as R
    from rust_lbfgs_amd.math import DeviceVec

    with R.Context(n) as ctx:
        vs = [DeviceVec(ctx) for _ in range(16)]
        for i, v in enumerate(vs):
            v.fill(1.0 + i)
        ctx.sync()

        def timed(pairs, reps=5):
            best = 1e9
            for _ in range(reps):
                ctx.sync()
                t0 = time.perf_counter()
                for a, b in pairs:
                    a.vecdot_slot(b, 20)
                ctx.sync()
                best = min(best, (time.perf_counter() - t0) / len(pairs))
            return best * 1e6

        cold = timed([(vs[2 * (i % 8)], vs[2 * (i % 8) + 1]) for i in range(40)])
        hot = timed([(vs[0], vs[1])] * 40)
        hot1 = timed([(vs[0], vs[0])] * 40)
        mb = n * 8 / 1e6
        print(f"n={n} ({mb:.0f} MB per vector), NT threshold {os.environ.get('LBFGS_HIP_NT_THRESHOLD_MB', 'default (128)')} MiB: "
              f"cold pair {cold:.1f} us = {2 * mb / cold:.2f} TB/s | same pair {hot:.1f} us = {2 * mb / hot:.2f} TB/s | "
              f"same vector twice {hot1:.1f} us = {2 * mb / hot1:.2f} TB/s requested", flush=True)
        for v in vs:
            v.free()


if __name__ == "__main__":
    if os.environ.get("MALL_PROBE_CHILD"):
        run(int(sys.argv[1]))
    else:
        for n in ([int(sys.argv[1])] if len(sys.argv) > 1 else [3_000_000, 6_250_000, 12_500_224, 25_000_192]):
            for nt in (None, "1"):
                env = dict(os.environ, MALL_PROBE_CHILD="1")
                if nt:
                    env["LBFGS_HIP_NT_THRESHOLD_MB"] = nt
                subprocess.run([sys.executable, os.path.abspath(__file__), str(n)], env=env, check=False)
