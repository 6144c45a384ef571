#!/usr/bin/env python3
"""Conditioning study of the vector-free (Gram) two-loop against the exact recursion.

For diagonal quadratics f = sum 0.5*a_i*x_i^2 - b_i*x_i with a_i = kappa^(u_i) (condition number kappa), the CPU
oracle (reference operation order) runs L-BFGS; at every iteration its history is handed to the device, which forms
the search direction twice -- exact fused recursion (lbfgs_hip_two_loop) and Gram-coefficient recursion
(lbfgs_hip_two_loop_gram) -- and both are compared with the oracle's own direction (step-locked).  Then whole runs
with and without with_vector_free(True) are compared (free-running).

    python tools/vector_free_study.py [--n 20000] [--iters 60]        (needs the GPU)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rust_lbfgs_amd import hotpath as H  # noqa: E402
from rust_lbfgs_amd.math import DeviceVec  # noqa: E402


def problem(n, kappa, seed):
    r = np.random.default_rng(seed)
    a = kappa ** r.random(n)
    a[0], a[-1] = 1.0, kappa
    b = 2.0 * r.random(n) - 1.0

    def evaluate(x, g):
        t = a * x
        g[:] = t - b
        return float(np.sum(x * (0.5 * t - b)))

    return evaluate


def rel(u, v):
    return float(np.max(np.abs(u - v)) / max(np.max(np.abs(v)), 1e-300))


def step_locked(n, m, kappa, iters):
    ev = problem(n, kappa, 1)
    st = O.lbfgs().with_m(m).with_epsilon(0.0).build(np.zeros(n), ev)
    worst_exact = worst_vf = 0.0
    g0 = gk = None
    done = 0
    with R.Context(n) as ctx:
        h1, h2 = H.History(ctx, m), H.History(ctx, m)
        gv, d1, d2 = DeviceVec(ctx), DeviceVec(ctx), DeviceVec(ctx)
        try:
            for _ in range(iters):
                end_before = st.end
                try:
                    p = st.propagate()
                except O.OracleError:
                    break  # converged to rounding: the line search cannot make progress
                if g0 is None:
                    g0 = p["gnorm"]
                gk = p["gnorm"]
                if p["niter"] == 1:
                    continue
                for h in (h1, h2):
                    h.s(end_before).upload(st.hist(end_before, "s"))
                    h.y(end_before).upload(st.hist(end_before, "y"))
                    h.set_scalars(ys=np.array([st.ys(j) for j in range(m)]))
                ctx.set_scalars(7, [st.gamma, 1.0])
                gv.upload(st.vec("gx"))
                h1.two_loop(d1, gv, st.k - 1, end_before, 7, 8, 12)
                h2.two_loop_gram(d2, gv, st.k - 1, end_before, 7, 8, 14)
                dref = st.vec("d")
                worst_exact = max(worst_exact, rel(d1.to_numpy(), dref))
                worst_vf = max(worst_vf, rel(d2.to_numpy(), dref))
                done += 1
        finally:
            st.close()
            h1.free(); h2.free(); gv.free(); d1.free(); d2.free()
    return done, worst_exact, worst_vf, gk / g0


def free_running(n, m, kappa, iters):
    ev = problem(n, kappa, 1)
    out = {}
    for vf in (False, True):
        x = np.zeros(n)
        rows = []
        try:
            R.lbfgs().with_m(m).with_epsilon(0.0).with_max_iterations(iters).with_vector_free(vf).minimize(
                x, ev, lambda p: rows.append((p.niter, p.neval, p.fx, p.gnorm)) and False)
        except R.LbfgsError:
            pass
        out[vf] = rows
    # the yardstick: the CPU oracle (sequential sums) free-running on the same problem.  Exact-on-GPU differs from it
    # only by summation order, so |exact - oracle| is the noise floor any correct implementation shows here.
    xo, rows = np.zeros(n), []
    try:
        O.lbfgs().with_m(m).with_epsilon(0.0).with_max_iterations(iters).minimize(
            xo, ev, lambda p: rows.append((p["niter"], p["neval"], p["fx"], p["gnorm"])) and False)
    except O.OracleError:
        pass
    out["oracle"] = rows
    k = min(len(out[False]), len(out[True]), len(rows))
    same_counts = all(out[False][i][:2] == out[True][i][:2] == rows[i][:2] for i in range(k))

    def fdiff(u, v):
        return max((abs(u[i][2] - v[i][2]) / max(abs(v[i][2]), 1e-300) for i in range(k)), default=0.0)

    return k, same_counts, fdiff(out[False], rows), fdiff(out[True], rows), fdiff(out[True], out[False])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--iters", type=int, default=60)
    a = ap.parse_args()
    print("| kappa | m | iterations | exact recursion vs oracle (max rel. err of d) | vector-free vs oracle | ‖g‖ reduction |"
          " free-running: same (niter, neval) in all three | max rel. diff of f: exact vs oracle | vector-free vs oracle | vector-free vs exact |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for kappa in (1e2, 1e4, 1e6, 1e8, 1e10):
        for m in (5, 10):
            it, we, wv, red = step_locked(a.n, m, kappa, a.iters)
            k, same, d_eo, d_vo, d_ve = free_running(a.n, m, kappa, a.iters)
            print(f"| {kappa:.0e} | {m} | {it} | {we:.1e} | {wv:.1e} | {red:.1e} | {same} ({k} it) | {d_eo:.1e} | {d_vo:.1e} | {d_ve:.1e} |",
                  flush=True)


if __name__ == "__main__":
    main()
