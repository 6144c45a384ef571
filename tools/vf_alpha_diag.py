#!/usr/bin/env python3
"""Diagnostic: per iteration of a step-locked run, the alphas of the oracle, of the exact device recursion and of the
vector-free one (tests/test_gpu_step_locked.py vector_free_row), slot by slot.   python tools/vf_alpha_diag.py [case]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rust_lbfgs_amd import hotpath as H  # noqa: E402
from rust_lbfgs_amd.math import DeviceVec  # noqa: E402
from tests import problems as P  # noqa: E402
from tests.test_gpu_step_locked import CASES  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "quadratic_m7"
n, m, tweak, oobj, dobj, x0kind, iters = CASES[case]
x = P.rosenbrock_x0(n) if x0kind == "rosenbrock" else np.zeros(n)
st = tweak(O.lbfgs().with_m(m)).build(x, oobj())
np.set_printoptions(precision=3, linewidth=250)
with R.Context(n) as ctx:
    hist = H.History(ctx, m)
    gv, dv = DeviceVec(ctx), DeviceVec(ctx)
    for _ in range(iters):
        if st.is_converged():
            break
        end_before = st.end
        try:
            p = st.propagate()
        except O.OracleError:
            break
        if p["niter"] == 1:
            continue
        hist.s(end_before).upload(st.hist(end_before, "s"))
        hist.y(end_before).upload(st.hist(end_before, "y"))
        ys = np.array([st.ys(j) for j in range(m)])
        gv.upload(st.vec("gx"))
        ctx.set_scalars(7, [st.gamma, 1.0])
        a_ref = np.array([st.alpha(j) for j in range(m)])
        hist.set_scalars(ys=ys, alpha=np.zeros(m))
        hist.two_loop_gram(dv, gv, st.k - 1, end_before, 7, 8, 24)
        dn2, gd, pred, cancel = ctx.scalars(24, 4)
        a_vf = hist.scalars()[1].copy()
        d_vf = dv.to_numpy()
        hist.set_scalars(ys=ys, alpha=np.zeros(m))
        hist.two_loop(dv, gv, st.k - 1, end_before, 7, 8, 12)
        a_ex = hist.scalars()[1].copy()
        d_ex = dv.to_numpy()
        dref = st.vec("d")
        sn = np.array([np.linalg.norm(st.hist(j, "s")) for j in range(m)])
        scale = sn * np.linalg.norm(st.vec("gx")) / np.abs(np.where(ys == 0, 1, ys))
        print(f"it {p['niter']:3d} gnorm {p['gnorm']:.2e} cancel {cancel:.2f} pred_err {abs(pred - dn2) / dn2:.1e} | d: vf {np.max(np.abs(d_vf - dref)) / np.max(np.abs(dref)):.1e} "
              f"exact {np.max(np.abs(d_ex - dref)) / np.max(np.abs(dref)):.1e} | alpha/max: vf {np.max(np.abs(a_vf - a_ref)) / np.max(np.abs(a_ref)):.1e} "
              f"exact {np.max(np.abs(a_ex - a_ref)) / np.max(np.abs(a_ref)):.1e}")
        print("      ref  ", a_ref)
        print("      vf-ref", a_vf - a_ref, " exact-ref", a_ex - a_ref)
        print("      |s||g|/ys", scale)
st.close()
