"""Step-locked parity at the strict bar of BASELINE.json: "per-iteration f, ||g|| and search direction
match the reference CPU path on IDENTICAL inputs within 1e-10 relative (f64)".

The CPU oracle runs the optimisation.  After each of its iterations the GPU is handed exactly the
oracle's inputs for that iteration -- the point x, the (s, y, ys) history, gamma -- and must
reproduce the oracle's outputs: f(x), ||g(x)||, the history pair just written, and the search
direction of the two-loop recursion (lbfgs.rs:569-604).  Nothing accumulates across iterations, so the
tolerance is the north star's 1e-10 with no allowance for trajectory drift.
"""
import numpy as np
import pytest

import rust_lbfgs_amd as R
from oracle import oracle as O
from rust_lbfgs_amd import hotpath as H, objectives
from rust_lbfgs_amd.math import DeviceVec
from tests import problems as P
from tests.test_gpu_parity import product_library, rel  # noqa: F401  (fixture + helper)

pytestmark = pytest.mark.gpu
RTOL = 1e-10

CASES = {
    # name: (n, m, builder tweaks, oracle objective, device objective, x0, iterations)
    "quadratic_m7": (4096, 7, lambda b: b.with_epsilon(0.0), O.quadratic, objectives.Quadratic, "zeros", 40),
    "quadratic_m10_big": (1_000_003, 10, lambda b: b.with_epsilon(0.0), O.quadratic, objectives.Quadratic, "zeros", 25),
    # 160 MB vectors: above the 128 MiB threshold, so every kernel runs in its streaming (`nt`) instantiation
    "quadratic_m3_streaming": (20_000_003, 3, lambda b: b.with_epsilon(0.0), O.quadratic, objectives.Quadratic, "zeros", 9),
    "rosenbrock_m10": (1000, 10, lambda b: b, O.rosenbrock, objectives.Rosenbrock, "rosenbrock", 30),
    "logistic_owlqn_m6": (4096, 6, lambda b: b.with_orthantwise(0.5, 0, None), O.logistic, objectives.Logistic, "zeros", 25),
    "logistic_owlqn_range": (5001, 6, lambda b: b.with_orthantwise(0.25, 100, 4000), O.logistic, objectives.Logistic,
                             "zeros", 25),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_step_locked(case):
    n, m, tweak, oobj, dobj, x0kind, iters = CASES[case]
    x = P.rosenbrock_x0(n) if x0kind == "rosenbrock" else np.zeros(n)
    b = tweak(O.lbfgs().with_m(m))
    owl = bool(b.param.orthantwise)
    st = b.build(x, oobj())
    with R.Context(n) as ctx:
        hist = H.History(ctx, m)
        xv, gv, pgv, dv, xpv, gpv = (DeviceVec(ctx) for _ in range(6))
        worst = dict(f=0.0, g=0.0, d=0.0, s=0.0, y=0.0, ys=0.0)
        done = 0
        for _ in range(iters):
            if st.is_converged():
                break
            end_before = st.end
            xp_h, gp_h = st.vec("x").copy(), st.vec("gx").copy()
            d_prev = st.vec("d").copy()  # the direction this iteration's line search moves along
            p = st.propagate()
            if p["niter"] == 1:
                continue
            done += 1
            x_h, g_h = st.vec("x"), st.vec("gx")
            # (1) f and ||g|| at the oracle's point
            xv.upload(x_h)
            H.objective_eval(dobj(), xv, gv, 0)
            f_dev = ctx.scalars(0)[0]
            f_ref, _ = O.eval_builtin(oobj(), np.ascontiguousarray(x_h))
            worst["f"] = max(worst["f"], abs(f_dev - f_ref) / abs(f_ref))
            if owl:
                c, s0, e0 = b.param.owl_c, b.param.owl_start, n if b.param.owl_end < 0 else min(b.param.owl_end, n)
                H.owlqn_post_eval(xv, gv, pgv, c, s0, e0, 2)
                l1, pgn2, _ = ctx.scalars(2, 3)
                worst["f"] = max(worst["f"], abs((f_dev + l1) - p["fx"]) / abs(p["fx"]))
                worst["g"] = max(worst["g"], abs(np.sqrt(pgn2) - p["gnorm"]) / p["gnorm"])
            else:
                worst["f"] = max(worst["f"], abs(f_dev - p["fx"]) / abs(p["fx"]))
                H.norms_sq(xv, gv, 14)
                worst["g"] = max(worst["g"], abs(np.sqrt(ctx.scalars(15)[0]) - p["gnorm"]) / p["gnorm"])
            # (2) the history pair written this iteration, from the oracle's (x, xp, g, gp)
            xpv.upload(xp_h); gpv.upload(gp_h); gv.upload(g_h)
            hist.update(end_before, xv, xpv, gv, gpv, p["step"], False, 6)
            worst["s"] = max(worst["s"], rel(hist.s(end_before).to_numpy(), st.hist(end_before, "s")))
            if not b.param.damping:
                worst["y"] = max(worst["y"], rel(hist.y(end_before).to_numpy(), st.hist(end_before, "y")))
            ys_dev = ctx.scalars(7)[0]
            worst["ys"] = max(worst["ys"], abs(ys_dev - st.ys(end_before)) / abs(st.ys(end_before)))
            # (2b) deferred trial points: the probe at the accepted step sees the oracle's f; the fused
            #      "accepted step + update" kernel reproduces x, g, s, y from (xp, d, step) alone
            if not owl and dobj in (objectives.Quadratic, objectives.Logistic):
                dv.upload(d_prev)
                H.objective_line_probe(dobj(), xpv, dv, p["step"], 20)
                worst["f"] = max(worst["f"], abs(ctx.scalars(20)[0] - p["fx"]) / abs(p["fx"]))
                x2, g2 = DeviceVec(ctx), DeviceVec(ctx)
                h2 = H.History(ctx, 1)
                h2.update_from_step(0, dobj(), x2, xpv, dv, p["step"], g2, gpv, p["step"], False, 30)
                assert np.array_equal(x2.to_numpy(), x_h)  # x = xp + step*d: the same two roundings as core.rs:157-158
                worst["s"] = max(worst["s"], rel(h2.s(0).to_numpy(), st.hist(end_before, "s")))
                worst["y"] = max(worst["y"], rel(h2.y(0).to_numpy(), st.hist(end_before, "y")))
                worst["ys"] = max(worst["ys"], abs(ctx.scalars(31)[0] - st.ys(end_before)) / abs(st.ys(end_before)))
                h2.free(); x2.free(); g2.free()
            # (3) search direction from the ORACLE's history (all m slots re-uploaded: identical inputs)
            for j in range(m):
                hist.s(j).upload(st.hist(j, "s")); hist.y(j).upload(st.hist(j, "y"))
            hist.set_scalars(ys=np.array([st.ys(j) for j in range(m)]), alpha=np.zeros(m))
            ctx.set_scalars(7, [st.gamma, 1.0])
            src = pgv if owl else gv
            if owl:
                pgv.upload(st.vec("pg"))
            new_end = hist.two_loop(dv, src, st.k - 1, end_before, 7, 8, 12)
            assert new_end == st.end
            if owl:
                H.constrain_direction(dv, pgv, s0, e0, 13)
            worst["d"] = max(worst["d"], rel(dv.to_numpy(), st.vec("d")))
        st.close()
        hist.free()
        for v in (xv, gv, pgv, dv, xpv, gpv):
            v.free()
    assert done >= min(10, iters - 1), done
    print(case, {k: f"{v:.2e}" for k, v in worst.items()})
    for k, v in worst.items():
        assert v <= RTOL, (case, k, v)
