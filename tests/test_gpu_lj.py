"""The device-resident Lennard-Jones evaluators (BASELINE config 5) against the oracle.

 * the reference's own LJ38 coordinates (examples/lj.rs:72-110, tests/golden/lj38_positions.json) through the exact
   all-pairs kernel and through the example's `main` (defaults, no damping) and config 5's damped run;
 * LJ_CELLS: the cutoff evaluator whose neighbour list the library builds on the device from a cell list and REBUILDS
   when atoms have moved -- checked, at every point a minimisation visits, against the oracle's statement of the same
   rule (a function of x alone: oracle_obj_lj_cells rebuilds its cells at every call), on runs in which atoms cross
   cell boundaries and the list is rebuilt several times.

The reference holds no expected value for its LJ example (it only prints), so the oracle's LJ is pinned by source
reading alone: parity unpinned, stated in DESIGN.md."""
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
ON_MOCK = __import__("os").environ.get("LBFGS_TEST_BACKEND") == "mock"


def test_lj38_reference_coordinates_allpairs():
    """examples/lj.rs:38-64,113-118 at the example's own start point: energy and gradient."""
    x = P.lj38_x0()
    fo, go = O.eval_builtin(O.lj(), x)
    with R.Context(len(x)) as ctx:
        xv, gv = DeviceVec(ctx, x), DeviceVec(ctx)
        H.objective_eval(objectives.LennardJones(), xv, gv, 0)
        f = ctx.scalars(0)[0]
        assert abs(f - fo) <= RTOL * abs(fo)
        assert rel(gv.to_numpy(), go) <= RTOL
        xv.free(); gv.free()
    assert fo < -100.0  # a bound cluster (the LJ38 global minimum is -173.93)


def test_lj38_example_main_defaults():
    """The example's `main` (lj.rs:112-131): lbfgs().minimize(positions, LJ closure, ..) with the crate defaults, the
    objective resident on the device.  Trajectory against the oracle; tolerance = the oracle's own sensitivity to the
    summation order where that exceeds 1e-10 (a free-running nonconvex run amplifies last-bit differences)."""
    x0 = P.lj38_x0()
    cfg = lambda b: b.with_max_iterations(40)
    fields = ("niter", "neval", "fx", "gnorm", "step")

    def oracle_run(mode):
        O.lib().oracle_set_dot_mode(mode)
        try:
            rows, x = [], x0.copy()
            cfg(O.lbfgs()).minimize(x, O.lj(), lambda p: rows.append(tuple(p[f] for f in fields)) and False)
            return rows
        finally:
            O.lib().oracle_set_dot_mode(0)

    rows_o, rows_w = oracle_run(0), oracle_run(1)
    rows_p, xp = [], x0.copy()
    cfg(R.lbfgs()).minimize(xp, objectives.LennardJones(),
                            lambda p: rows_p.append((p.niter, p.neval, p.fx, p.gnorm, p.step)) and False)
    assert len(rows_p) == len(rows_o) >= 10
    floor = 0.0
    for a, w, b in zip(rows_o, rows_w, rows_p):
        if a[:2] != w[:2]:
            break
        floor = max(floor, max(abs(u - v) / max(abs(u), 1e-3) for u, v in zip(a[2:], w[2:])))
        assert a[:2] == b[:2]
        for u, v in zip(a[2:], b[2:]):
            assert abs(u - v) <= max(1e-10, 50.0 * floor) * max(abs(u), 1e-3), (a, b, floor)
    assert rows_p[-1][2] < rows_p[0][2]


# ------------------------------------------------------------------------------------------------ LJ_CELLS
def _jittered_lattice(nside, spacing, jitter, seed):
    g = np.stack(np.meshgrid(*[np.arange(nside)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    return (g * spacing + np.random.default_rng(seed).uniform(-jitter, jitter, g.shape)).reshape(-1)


def _check_point(ctx, obj, xv, gv, x, rc):
    H.objective_eval(obj, xv.upload(x), gv, 0)
    f = ctx.scalars(0)[0]
    fo, go = O.eval_builtin(O.lj_cells(rc), np.ascontiguousarray(x))
    assert abs(f - fo) <= RTOL * max(abs(fo), 1.0), (f, fo)
    assert rel(gv.to_numpy(), go) <= RTOL
    return f


@pytest.mark.parametrize("nside,rc,skin", [(3, 2.5, 0.3), (9, 2.5, 0.3), (12, 1.8, 0.5)])
def test_lj_cells_matches_the_cutoff_rule_through_rebuilds(nside, rc, skin):
    """Evaluate, move every atom a little (list kept), move some atoms far (list rebuilt), shuffle the whole system
    (everything changes cell): each result is the cutoff sum at that x."""
    rng = np.random.default_rng(nside)
    x = _jittered_lattice(nside, 1.15, 0.07, nside)
    n = len(x)
    obj = objectives.LennardJonesCells(rc, skin)
    with R.Context(n) as ctx:
        xv, gv = DeviceVec(ctx), DeviceVec(ctx)
        _check_point(ctx, obj, xv, gv, x, rc)
        r0 = ctx.lj_cells_stats()
        x1 = x + rng.uniform(-0.2, 0.2, n) * skin / np.sqrt(3.0) / 2.0   # every displacement < skin/2: same list
        _check_point(ctx, obj, xv, gv, x1, rc)
        r1 = ctx.lj_cells_stats()
        x2 = x1.copy()
        x2[: 3 * (n // 30 + 1)] += 0.9 * skin                               # a few atoms leave their skin/2 sphere
        _check_point(ctx, obj, xv, gv, x2, rc)
        r2 = ctx.lj_cells_stats()
        perm = rng.permutation(n // 3)
        x3 = x2.reshape(-1, 3)[perm].reshape(-1)                             # atom i sits where atom perm[i] was
        _check_point(ctx, obj, xv, gv, x3, rc)
        r3 = ctx.lj_cells_stats()
        # bitwise determinism of list and sums: the same point again, after a detour that forces a rebuild
        H.objective_eval(obj, xv.upload(x3), gv, 0)
        f_a, g_a = ctx.scalars(0)[0], gv.to_numpy()
        _check_point(ctx, obj, xv, gv, x, rc)
        H.objective_eval(obj, xv.upload(x3), gv, 0)
        assert ctx.scalars(0)[0] == f_a and np.array_equal(gv.to_numpy(), g_a)
        xv.free(); gv.free()
    if not ON_MOCK:
        assert r0[0] == 1 and r0[1] == 1                    # first use builds
        assert r1[0] == 1 and r1[1] == 2                    # small moves: no rebuild
        assert r2[0] == 2 and r2[1] == 4                    # rebuild + re-evaluation
        assert r3[0] == 3
        assert 4 <= r3[2] <= 128 and r3[2] % 4 == 0


def test_lj_cells_minimisation_with_atoms_crossing_cells():
    """Damped L-BFGS (BASELINE config 5's optimiser settings) on a jittered simple-cubic block: the block relaxes
    substantially, atoms cross cell boundaries and the list is rebuilt along the way.  At EVERY iterate the device's
    f and g are compared with the oracle's cutoff rule at the same x (step-locked: 1e-10 flat), the run must lower the
    energy, and a second run must be bitwise identical."""
    nside, rc, skin = 8, 2.5, 0.3
    x0 = _jittered_lattice(nside, 1.25, 0.12, 21)
    obj = objectives.LennardJonesCells(rc, skin)
    rl = rc + skin

    def run():
        pts = []
        x = x0.copy()
        with R.Context(len(x)) as ctx:
            b = R.lbfgs().with_damping(True).with_max_iterations(120).with_epsilon(1e-9)
            rep = b.minimize(x, obj, lambda p: pts.append((p.fx, p.x, p.gx)) and False, ctx=ctx)
            stats = ctx.lj_cells_stats()
        return rep, x, pts, stats

    rep, x, pts, stats = run()
    assert len(pts) >= 60
    for fx, px, pg in pts[::3] + pts[-1:]:
        fo, go = O.eval_builtin(O.lj_cells(rc), np.ascontiguousarray(px))
        assert abs(fx - fo) <= RTOL * abs(fo), (fx, fo)
        assert rel(pg, go) <= RTOL
    assert pts[-1][0] < pts[0][0] - 50.0
    moved = np.linalg.norm((x - x0).reshape(-1, 3), axis=1)
    lo = np.minimum(x0.reshape(-1, 3).min(0), x.reshape(-1, 3).min(0))
    crossed = np.any(np.floor((x0.reshape(-1, 3) - lo) / rl) != np.floor((x.reshape(-1, 3) - lo) / rl), axis=1)
    assert moved.max() > skin and crossed.sum() >= 5, (moved.max(), crossed.sum())
    if not ON_MOCK:
        assert stats[0] >= 3, stats                      # the list was rebuilt while the atoms moved
        assert stats[1] > stats[0]                       # ... and most evaluations reused it
    rep2, x2, pts2, stats2 = run()
    assert np.array_equal(x, x2) and rep.fx == rep2.fx and stats == stats2


def test_lj_cells_errors():
    x = _jittered_lattice(5, 1.1, 0.05, 1)
    with R.Context(len(x)) as ctx:
        xv, gv = DeviceVec(ctx, x), DeviceVec(ctx)
        if not ON_MOCK:
            with pytest.raises(R.LbfgsError) as e:       # 4 list entries per atom cannot hold ~60 neighbours
                H.objective_eval(objectives.LennardJonesCells(2.5, 0.3, max_nbr=4), xv, gv, 0)
            assert "max_nbr" in str(e.value)
            with pytest.raises(R.LbfgsError):
                H.objective_eval(objectives.LennardJonesCells(2.5, 0.0), xv, gv, 0)   # no skin
        bad = x.copy()
        bad[7] = np.nan
        with pytest.raises(R.LbfgsError):
            H.objective_eval(objectives.LennardJonesCells(2.5, 0.3), xv.upload(bad), gv, 0)
        # ... and the context is still usable afterwards
        H.objective_eval(objectives.LennardJonesCells(2.5, 0.3), xv.upload(x), gv, 0)
        fo, _ = O.eval_builtin(O.lj_cells(2.5), x)
        assert abs(ctx.scalars(0)[0] - fo) <= RTOL * abs(fo)
        xv.free(); gv.free()


def test_lj_cells_single_precision_list_gives_the_same_bits(monkeypatch):
    """The list is built from single-precision candidate tests with a safety margin (a slight superset of the
    double-precision list: lj.h).  The evaluation tests every entry against the cutoff in double precision, so energy
    and gradient must be BIT-identical between the two builds, at every point of a run with rebuilds -- also for a block
    far from the origin (large coordinates are what single precision is worst at)."""
    if ON_MOCK:
        pytest.skip("a property of the HIP kernels")
    rc, skin = 2.5, 0.3
    obj = objectives.LennardJonesCells(rc, skin)
    rng = np.random.default_rng(11)
    for shift in (0.0, 3.0e3):
        x0 = _jittered_lattice(9, 1.15, 0.08, 4) + shift
        pts = [x0, x0 + rng.uniform(-0.05, 0.05, len(x0)), x0 + rng.uniform(-0.4, 0.4, len(x0)),
               (x0.reshape(-1, 3)[rng.permutation(len(x0) // 3)]).reshape(-1)]
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("LBFGS_HIP_LJ_BUILD_FP32", mode)
            out = []
            with R.Context(len(x0)) as ctx:
                xv, gv = DeviceVec(ctx), DeviceVec(ctx)
                for p in pts:
                    H.objective_eval(obj, xv.upload(p), gv, 0)
                    out.append((ctx.scalars(0)[0], gv.to_numpy()))
                out.append(ctx.lj_cells_stats())
                xv.free(); gv.free()
            res[mode] = out
        for a, b in zip(res["1"][:-1], res["0"][:-1]):
            assert a[0] == b[0] and np.array_equal(a[1], b[1])
        assert res["1"][-1][:2] == res["0"][-1][:2]                  # the same rebuilds and evaluations
        assert res["1"][-1][2] >= res["0"][-1][2]                    # the single-precision list is a superset
        fo, go = O.eval_builtin(O.lj_cells(rc), np.ascontiguousarray(pts[2]))
        assert abs(res["1"][2][0] - fo) <= RTOL * abs(fo) and rel(res["1"][2][1], go) <= RTOL


def test_lj_cells_trial_forms_its_point_and_sums_gd_in_the_evaluation(monkeypatch):
    """A line-search trial of LJ_CELLS (lbfgs_hip_objective_line_eval: take_line_step + evaluate + dg_unchecked,
    core.rs:155-158,119-121,114-116): the trial point is formed by the pass that checks the neighbour list and g.d is summed
    by the kernel that forms g -- against the three-launch sequence the other Lennard-Jones evaluators take
    (LBFGS_HIP_LJ_FUSED_TRIAL=0): x, g and f bit for bit, g.d to rounding; for a step that keeps the list, a step that makes
    it stale (rebuild in the middle of the trial) and a trial before anything was evaluated (the list is built at the trial
    point)."""
    if ON_MOCK:
        pytest.skip("a property of the HIP kernels")
    rc, skin = 2.5, 0.3
    obj = objectives.LennardJonesCells(rc, skin)
    xp_h = _jittered_lattice(9, 1.15, 0.08, 31)
    n = len(xp_h)
    d_h = np.random.default_rng(5).standard_normal(n)
    d_h *= 1.0 / np.max(np.abs(d_h))
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LBFGS_HIP_LJ_FUSED_TRIAL", mode)
        out = []
        with R.Context(n) as ctx:
            x, xp, d, g = DeviceVec(ctx), DeviceVec(ctx, xp_h), DeviceVec(ctx, d_h), DeviceVec(ctx)
            for t in (0.31 * skin, 0.02 * skin, 1.7 * skin, 1.6 * skin, 0.4 * skin):   # no list yet; kept; stale; kept; stale again
                H.objective_line_eval(obj, x, xp, d, t, g, 20)
                f, gd = ctx.scalars(20, 2)
                out.append((t, f, gd, x.to_numpy(), g.to_numpy()))
            out.append(ctx.lj_cells_stats())
            for v in (x, xp, d, g):
                v.free()
        res[mode] = out
    # the same builds either way: the first trial's, the stale one's, and the last trial's (1.3 skin back from where the list was built)
    assert res["1"][-1][:2] == res["0"][-1][:2] and res["1"][-1][0] == 3
    for a, b in zip(res["1"][:-1], res["0"][:-1]):
        t = a[0]
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[3], xp_h + t * d_h)   # x = xp + t*d, two roundings
        assert a[1] == b[1] and np.array_equal(a[4], b[4])
        assert abs(a[2] - b[2]) <= 1e-12 * abs(b[2])
        fo, go = O.eval_builtin(O.lj_cells(rc), np.ascontiguousarray(a[3]))
        assert abs(a[1] - fo) <= RTOL * abs(fo) and rel(a[4], go) <= RTOL
        assert abs(a[2] - O.vecdot(go, d_h)) <= RTOL * abs(O.vecdot(go, d_h))
