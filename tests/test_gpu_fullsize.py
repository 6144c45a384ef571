"""Full-size (BASELINE.json: n = 1e8, m = 10) checks through size-independent properties, all on the device:
the oracle cannot run at this size in seconds, so parity is carried by (a) the fused two-loop against the
reference's UNFUSED sequence of primitives on the same device data, (b) linearity of the recursion,
(c) bitwise determinism, (d) exact integer-valued sums, and (e) `nt`-hinted kernels == plain kernels."""
import os

import numpy as np
import pytest

import rust_lbfgs_amd as R
from rust_lbfgs_amd import hotpath as H, objectives
from rust_lbfgs_amd.math import DeviceVec
from tests.test_gpu_parity import product_library  # noqa: F401

pytestmark = pytest.mark.gpu

N_FULL = int(os.environ.get("LBFGS_TEST_FULL_N", 100_000_000))
M = 10


def _fill_history(ctx, hist, tmp):
    """Deterministic, well-conditioned (s_j, y_j): s from the hashed quadratic's gradient at a constant
    point, y = s scaled by hashed positive curvatures (so y.s > 0), all generated on the device."""
    q = objectives.Quadratic()
    for j in range(hist.m):
        tmp.fill(0.25 + 0.1 * j)
        H.objective_eval(q, tmp, hist.s(j), 0)                 # s_j = a*c_j - b
        H.objective_eval(q, hist.s(j), hist.y(j), 0)           # y_j = a*s_j - b
        hist.y(j).vecadd(hist.s(j), 2.0)                       # keep y.s comfortably positive
    ys = [hist.y(j).vecdot(hist.s(j)) for j in range(hist.m)]
    assert all(v > 0 for v in ys)
    hist.set_scalars(ys=np.array(ys), alpha=np.zeros(hist.m))
    return ys


@pytest.mark.parametrize("n", [N_FULL, 12_500_003])
def test_two_loop_properties_at_full_size(n):
    with R.Context(n) as ctx:
        hist = H.History(ctx, M)
        g, d1, d2, tmp = (DeviceVec(ctx) for _ in range(4))
        ys = _fill_history(ctx, hist, tmp)
        tmp.fill(-0.3)
        H.objective_eval(objectives.Logistic(), tmp, g, 0)     # some non-trivial g
        yy = hist.y(3).vecdot(hist.y(3))
        ctx.set_scalars(7, [ys[3], yy])
        k, end = 37, 3                                          # bound = m = 10, ring wrapped
        # (a) fused (8b passes) vs the reference's unfused primitive sequence (10b+2 passes)
        ne1 = hist.two_loop(d1, g, k, end, 7, 8, 12)
        dn2 = ctx.scalars(12)[0]
        _, alpha_fused = hist.scalars()
        d2.vecncpy(g)
        ne2 = hist.two_loop_unfused(d2, k, end, 7, 8)
        _, alpha_unfused = hist.scalars()
        assert ne1 == ne2 == 4
        tmp.vecdiff(d1, d2)
        dnorm = d1.vec2norm()
        assert dnorm > 0 and np.isfinite(dnorm)
        assert tmp.vec2norm() <= 1e-12 * dnorm
        assert abs(dn2 - dnorm * dnorm) <= 1e-12 * dn2
        assert np.max(np.abs(alpha_fused - alpha_unfused)) <= 1e-12 * np.max(np.abs(alpha_unfused))
        # (c) determinism: the same launch twice is bitwise identical
        hist.two_loop(d2, g, k, end, 7, 8, 12)
        tmp.vecdiff(d1, d2)
        assert tmp.vec2norm() == 0.0
        assert ctx.scalars(12)[0] == dn2
        # (b) linearity: H(4g) == 4 H(g) exactly (scaling by a power of two commutes with every rounding)
        g.vecscale(4.0)
        hist.two_loop(d2, g, k, end, 7, 8, 12)
        d1.vecscale(4.0)
        tmp.vecdiff(d1, d2)
        assert tmp.vec2norm() == 0.0
        # (d) integer-valued sums are exact in any order
        d1.fill(1.0); d2.fill(3.0)
        assert d1.vecdot(d2) == 3.0 * n
        assert d2.vec2norm() == np.sqrt(9.0 * n)
        hist.free()
        for v in (g, d1, d2, tmp):
            v.free()


def test_streaming_hint_kernels_equal_plain_kernels(monkeypatch):
    """From 128 MiB per vector the library launches the `nt`-hinted instantiations, from 64 MiB the ones with `nt` on the
    stores only.  Force each at a small size and require bit-identical results to the plain ones (same arithmetic,
    same order)."""
    n = 3_000_017
    out = {}
    for mode, thr, thr_st in (("plain", "1000000", "1000000"), ("nt", "0", "0"), ("nt_stores", "1000000", "0")):
        monkeypatch.setenv("LBFGS_HIP_NT_THRESHOLD_MB", thr)
        monkeypatch.setenv("LBFGS_HIP_NT_STORE_THRESHOLD_MB", thr_st)
        with R.Context(n) as ctx:
            hist = H.History(ctx, 4)
            g, d, tmp, x, xp = (DeviceVec(ctx) for _ in range(5))
            q = objectives.Quadratic()
            for j in range(4):
                tmp.fill(0.5 + j)
                H.objective_eval(q, tmp, hist.s(j), 0)
                H.objective_eval(q, hist.s(j), hist.y(j), 0)
                hist.y(j).vecadd(hist.s(j), 2.0)
            ys = [hist.y(j).vecdot(hist.s(j)) for j in range(4)]
            hist.set_scalars(ys=np.array(ys))
            ctx.set_scalars(7, [ys[1], hist.y(1).vecdot(hist.y(1))])
            tmp.fill(0.1)
            H.objective_eval(objectives.Logistic(), tmp, g, 0)
            hist.two_loop(d, g, 9, 1, 7, 8, 12)
            xp.fill(0.2)
            H.objective_line_eval(q, x, xp, d, 1e-3, g, 0)
            hist.update(2, x, xp, g, d, 0.5, True, 6)
            out[mode] = (d.to_numpy(), x.to_numpy(), g.to_numpy(), hist.s(2).to_numpy(), hist.y(2).to_numpy(),
                         ctx.scalars(0, 13))
            hist.free()
            for v in (g, d, tmp, x, xp):
                v.free()
    for other in ("nt", "nt_stores"):
        for a, b in zip(out["plain"], out[other]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("n_default", [10_000_000, 20_000_001], ids=["config3", "twice_config3_hybrid_two_loop"])
def test_owlqn_properties_at_config3_size(n_default):
    """BASELINE.json config 3 at its own size (OWL-QN, L1 logistic, n = 1e7, m = 6, c = 0.5), through size-independent
    properties on the device:
      (a) the fused trial kernel (projected line step + evaluate + x1norm + pseudo-gradient + g.d in one pass,
          line.rs:740-743 over core.rs:155-164,119-126) == the unfused sequence of the four primitives: x, g, pg
          BITWISE, the five sums to 1e-12;
      (b) two_loop_owlqn (orthant projection folded into the last step) == two_loop + constrain_direction
          (orthantwise.rs:140-161): d bitwise, ||d||^2 and pg.d to 1e-12;
      (c) bitwise determinism of both;
      (d) coordinates whose |g| <= c at x = 0 stay exactly 0 (the reason config 3 exercises the projection).
    Also at twice that size (odd n), where the two-loop is the HYBRID persistent kernel: part of q on the chip, the rest
    streamed, the projection applied in both parts."""
    n, m, c = int(os.environ.get("LBFGS_TEST_CONFIG3_N", n_default)), 6, 0.5
    q = objectives.Logistic()
    with R.Context(n) as ctx:
        hist = H.History(ctx, m)
        xp, gp, pg, wp, d, tmp = (DeviceVec(ctx) for _ in range(6))
        x1, g1, pg1, x2, g2, pg2 = (DeviceVec(ctx) for _ in range(6))
        close = lambda a, b: abs(a - b) <= 1e-12 * max(abs(a), abs(b), 1e-300)
        same = lambda u, v: (tmp.vecdiff(u, v), tmp.vec2norm())[1] == 0.0

        def trial_pair(t):
            H.objective_owlqn_line_eval(q, x1, xp, d, t, wp, g1, pg1, c, 0, n, 20)
            fused = ctx.scalars(20, 5)                      # f, g.d, c*sum|x|, ||pg||^2, ||x||^2
            H.line_step(x2, xp, d, t, wp, 0, n)
            H.objective_eval(q, x2, g2, 30)
            H.owlqn_post_eval(x2, g2, pg2, c, 0, n, 32)     # c*sum|x|, ||pg||^2, ||x||^2
            g2.vecdot_slot(d, 31)
            unfused = ctx.scalars(30, 5)
            assert same(x1, x2) and same(g1, g2) and same(pg1, pg2)
            for a, b in zip(fused, unfused):
                assert close(a, b), (fused, unfused)
            H.objective_owlqn_line_eval(q, x1, xp, d, t, wp, g1, pg1, c, 0, n, 20)  # (c) determinism
            assert np.array_equal(ctx.scalars(20, 5), fused) and same(x1, x2) and same(pg1, pg2)
            return fused

        # round 1 from x = 0 (every coordinate takes the x == 0 branch of orthantwise.rs:95-110)
        xp.fill(0.0)
        H.objective_eval(q, xp, gp, 0)
        H.owlqn_post_eval(xp, gp, pg, c, 0, n, 2)
        H.orthant_select(wp, xp, pg)
        d.vecncpy(pg)
        s1 = trial_pair(0.75)
        zeros = n - int(np.count_nonzero(x1.to_numpy()))
        assert 0.3 * n < zeros < 0.37 * n                   # a_i <= 1 <=> |g_i(0)| <= c: a third of the coordinates
        # round 2 from that point (mixed zero / non-zero coordinates: both branches, and real sign flips)
        xp.veccpy(x1); gp.veccpy(g1); pg.veccpy(pg1)
        H.orthant_select(wp, xp, pg)
        d.vecncpy(pg)
        s2 = trial_pair(1.0)
        assert s2[0] + s2[2] < s1[0] + s1[2] < n * np.log(2.0)  # the objective f + c|x|_1 went down twice
        # (b) the two forms of the OWL-QN two-loop
        _fill_history(ctx, hist, tmp)
        ys3 = hist.y(3).vecdot(hist.s(3))
        ctx.set_scalars(7, [ys3, hist.y(3).vecdot(hist.y(3))])
        ne1 = hist.two_loop_owlqn(x1, pg, 23, 3, 0, n, 7, 8, 13)
        a = ctx.scalars(13, 4)                              # ||d||^2 before (slot 14 unused); ||d||^2, pg.d after the projection
        ne2 = hist.two_loop(x2, pg, 23, 3, 7, 8, 40)
        b_pre = ctx.scalars(40, 2)
        H.constrain_direction(x2, pg, 0, n, 42)
        b_post = ctx.scalars(42, 2)
        assert ne1 == ne2 == 4 and same(x1, x2)
        for u, v in zip((a[0], a[2], a[3]), (b_pre[0], b_post[0], b_post[1])):
            assert close(u, v), (a, b_pre, b_post)
        assert a[2] > 0 and a[3] < 0                        # a descent direction survives the projection
        hist.two_loop_owlqn(g1, pg, 23, 3, 0, n, 7, 8, 13)
        assert same(g1, x1) and np.array_equal(ctx.scalars(13, 4), a)
        hist.free()
        for v in (xp, gp, pg, wp, d, tmp, x1, g1, pg1, x2, g2, pg2):
            v.free()


def test_lj_cells_properties_at_config5_size():
    """BASELINE.json config 5 at its own size (Lennard-Jones, 1e6 atoms, n = 3e6) through the library-built,
    rebuildable neighbour list (LJ_CELLS): energy and forces against the oracle's cutoff rule (its cell list is O(N), a
    few seconds here), bitwise determinism across a forced rebuild, Newton's third law (the forces sum to zero), and
    the rebuild trigger at this size."""
    nside = int(os.environ.get("LBFGS_TEST_CONFIG5_NSIDE", 100))
    rc, skin = 2.5, 0.3
    rng = np.random.default_rng(5)
    g3 = np.stack(np.meshgrid(*[np.arange(nside, dtype=np.float64)] * 3, indexing="ij"), -1).reshape(-1, 3)
    x = (g3 * 1.1 + rng.uniform(-0.06, 0.06, g3.shape)).reshape(-1)
    del g3
    n = len(x)
    obj = objectives.LennardJonesCells(rc, skin)
    from oracle import oracle as O

    with R.Context(n) as ctx:
        xv, gv = DeviceVec(ctx, x), DeviceVec(ctx)
        H.objective_eval(obj, xv, gv, 0)
        f1, g1 = ctx.scalars(0)[0], gv.to_numpy()
        fo, go = O.eval_builtin(O.lj_cells(rc), x)
        assert abs(f1 - fo) <= 1e-10 * abs(fo)
        assert np.max(np.abs(g1 - go)) <= 1e-10 * np.max(np.abs(go))
        for k in range(3):  # Newton's third law: every pair term enters two atoms with opposite signs
            assert abs(np.sum(g1[k::3])) <= 1e-9 * np.sum(np.abs(g1[k::3]))
        # a displaced copy forces a rebuild; coming back must reproduce the first result bit for bit
        x2 = x.copy()
        x2[: 3 * 1000] += 0.8 * skin
        H.objective_eval(obj, xv.upload(x2), gv, 0)
        f2 = ctx.scalars(0)[0]
        assert f2 != f1
        H.objective_eval(obj, xv.upload(x), gv, 0)
        assert ctx.scalars(0)[0] == f1 and np.array_equal(gv.to_numpy(), g1)
        if os.environ.get("LBFGS_TEST_BACKEND") != "mock":
            rebuilds, evals, longest = ctx.lj_cells_stats()
            assert rebuilds == 3 and evals == 5 and 40 <= longest <= 128
        xv.free(); gv.free()
