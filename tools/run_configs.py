#!/usr/bin/env python3
"""Measure BASELINE.json's single-GPU configs (2, 3, the host-closure variant of 2, and 4 at P=1) with the
CPU oracle timed beside each on a bounded sample.  Prints one JSON object per config.

    python tools/run_configs.py [--quick]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rust_lbfgs_amd import _ffi, objectives  # noqa: E402


def gpu_run(n, builder, evaluate, warm, timed, m):
    ctx = R.Context(n)
    st = builder.build(np.zeros(n), evaluate, ctx=ctx)
    rows = []
    for _ in range(warm):
        if st.is_converged():
            break
        rows.append(st.propagate())
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.prof_enable(False)
    ctx.sync()
    t0 = time.perf_counter()
    done = 0
    trials = 0
    for it in range(timed):
        if st.is_converged():
            break
        ctx.prof_enable(it % 5 == 0)  # sampled kernel timing, as bench.py
        p = st.propagate()
        trials += p.ncall
        done += 1
    ctx.sync()
    dt = time.perf_counter() - t0
    ntl, ms_tl = ctx.prof_read(_ffi.K_TWOLOOP_ALL)
    nst, ms_st = ctx.prof_read(_ffi.K_TWOLOOP_STEP)
    nres, _ = ctx.prof_read(_ffi.K_TWOLOOP_RESIDENT)
    n_res = min(ctx.resident_elements(), n) if nres else 0
    rep = st.report()
    st.close()
    ctx.close()
    out = dict(iters=done, iters_per_sec=done / dt, ms_per_iter=dt / max(done, 1) * 1e3, trials_per_iter=trials / max(done, 1),
               fx=rep.fx, gnorm=rep.gnorm)
    if ntl:
        t = ms_tl / ntl
        # bytes the recursion has to move: 8m passes of an n-vector with a kernel per step; the persistent kernel
        # keeps n_res elements of the running vector on the chip, which cost 4m+1 passes instead
        nbytes = 8.0 * ((4 * m + 1) * n_res + 8 * m * (n - n_res)) if n_res else 64.0 * m * n
        out.update(two_loop_ms=t, two_loop_bytes=nbytes, two_loop_GBps=nbytes / (t * 1e-3) / 1e9,
                   two_loop_frac=nbytes / (t * 1e-3) / 8e12, two_loop_resident_elements=n_res)
    if nst:
        t = ms_st / nst
        out.update(step_kernel_ms=t, step_kernel_GBps=32.0 * n / (t * 1e-3) / 1e9)
    return out


def cpu_run(n, builder, evaluate, warm, timed):
    st = builder.build(np.zeros(n), evaluate)
    for _ in range(warm):
        st.propagate()
    t0 = time.perf_counter()
    for _ in range(timed):
        st.propagate()
    dt = time.perf_counter() - t0
    st.close()
    return dict(n_sample=n, iters_per_sec_at_sample=timed / dt, cores=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    scale = 10 if a.quick else 1
    cfgs = []
    n2 = 10_000_000 // scale
    cfgs.append(("config2_quadratic_n1e7_m7_morethuente", n2, 7,
                 lambda B: B.lbfgs().with_m(7).with_epsilon(0.0), objectives.Quadratic(), O.quadratic(), 10, 50, 2_000_000))
    cfgs.append(("config3_owlqn_logistic_n1e7_m6", n2, 6,
                 lambda B: B.lbfgs().with_orthantwise(0.5, 0, None).with_epsilon(0.0), objectives.Logistic(), O.logistic(),
                 8, 14, 1_000_000))
    cfgs.append(("config4_quadratic_n1e8_m10_P1", 100_000_000 // scale, 10,
                 lambda B: B.lbfgs().with_m(10).with_epsilon(0.0), objectives.Quadratic(), O.quadratic(), 12, 50, 4_000_000))
    for name, n, m, mk, dev, orc, warm, timed, n_cpu in cfgs:
        if a.only and a.only not in name:
            continue
        g = gpu_run(n, mk(R), dev, warm, timed, m)
        c = cpu_run(n_cpu // scale, mk(O), orc, warm, max(4, timed // 8))
        c["iters_per_sec_scaled_to_n"] = c["iters_per_sec_at_sample"] * c["n_sample"] / n
        print(json.dumps(dict(config=name, n=n, m=m, gpu=g, cpu_oracle_1core=c)), flush=True)
    # config 5: damped L-BFGS on a Lennard-Jones system of 1e6 atoms (n = 3e6).  examples/lj.rs is all-pairs O(N^2):
    # at 1e6 atoms that is 5e11 pairs per evaluation, so the cutoff evaluator is used: the same pair terms within
    # 2.5 sigma, energy shifted by v(rc) (documented deviation), through the neighbour list the library
    # builds on the device from a cell list and rebuilds as the atoms move (LJ_CELLS, skin 0.3 sigma)
    if not a.only or "config5" in a.only:
        nside = 100 // (2 if a.quick else 1)
        g3 = np.stack(np.meshgrid(*[np.arange(nside, dtype=np.float64)] * 3, indexing="ij"), -1).reshape(-1, 3)
        x0 = (g3 * 1.12 + np.random.default_rng(5).uniform(-0.03, 0.03, g3.shape)).reshape(-1)
        n = len(x0)
        for label, warm, timed in (("first_35_iterations", 5, 30), ("iterations_6_to_305", 5, 300)):
            ctx = R.Context(n)
            st = R.lbfgs().with_damping(True).with_epsilon(0.0).build(x0, objectives.LennardJonesCells(2.5, 0.3), ctx=ctx)
            for _ in range(warm):
                st.propagate()
            r0 = ctx.lj_cells_stats()
            ctx.prof_enable(True); ctx.prof_reset(); ctx.sync()
            t0 = time.perf_counter()
            done = trials = 0
            for _ in range(timed):
                p = st.propagate(); trials += p.ncall; done += 1
            ctx.sync()
            dt = time.perf_counter() - t0
            nev, ms_ev = ctx.prof_read(_ffi.K_EVAL)
            ntl, ms_tl = ctx.prof_read(_ffi.K_TWOLOOP_ALL)
            r1 = ctx.lj_cells_stats()
            x1 = st.download("x")
            rep = st.report(); st.close(); ctx.close()
            print(json.dumps(dict(config="config5_lj_damped_1e6_atoms_cell_list", window=label, n=n, m=6, natoms=n // 3,
                                  cutoff=2.5, skin=0.3, longest_neighbour_list=r1[2],
                                  list_rebuilds_in_window=r1[0] - r0[0], evaluations_in_window=r1[1] - r0[1],
                                  max_displacement_from_start=float(np.max(np.linalg.norm((x1 - x0).reshape(-1, 3), axis=1))),
                                  gpu=dict(iters=done, iters_per_sec=done / dt, ms_per_iter=dt / done * 1e3,
                                           trials_per_iter=trials / done, lj_eval_ms_incl_rebuilds=ms_ev / max(nev, 1),
                                           two_loop_ms=ms_tl / max(ntl, 1), fx=rep.fx, gnorm=rep.gnorm))), flush=True)
    # config 2 through the DROP-IN host closure: x and g cross PCIe on every evaluate
    if not a.only or "closure" in a.only:
        n = n2
        idx = np.arange(n, dtype=np.float64)
        av = 1.0 + 999.0 * ((idx * 0.6180339887498949) % 1.0) ** 2
        bv = 2.0 * ((idx * 0.7548776662466927) % 1.0) - 1.0

        def ev(x, g):
            t = av * x
            np.subtract(t, bv, out=g)
            return float(np.dot(x, 0.5 * t - bv))

        g = gpu_run(n, R.lbfgs().with_m(7).with_epsilon(0.0), ev, 5, 15, 7)
        print(json.dumps(dict(config="config2_via_host_closure_pcie_inclusive", n=n, m=7, gpu=g,
                              note="numpy objective on the host; x downloaded and g uploaded on every evaluate")), flush=True)


if __name__ == "__main__":
    main()
