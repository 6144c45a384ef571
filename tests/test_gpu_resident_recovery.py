"""The chip-wide resident two-loop kernel (rust-lbfgs_amd/csrc/resident.h) assumes that it gets every CU it asks for.
When it does not -- another resident kernel on the GPU, a CU-masked queue -- its hand-offs time out.  That must cost a
wait, never the run: `lbfgs_hip_scalars_read` re-runs that recursion with a kernel per step (its inputs are intact) and
the context stays on that path.  Results are checked against the CPU oracle (lbfgs.rs:569-604 as restated there)."""
import os
import threading

import numpy as np
import pytest

import rust_lbfgs_amd as R
from oracle import oracle as O
from rust_lbfgs_amd import objectives
from tests.test_gpu_parity import product_library  # noqa: F401

pytestmark = pytest.mark.gpu


def run_oracle(n, m, iters, owl=None):
    rows, x = [], np.zeros(n)
    b = O.lbfgs().with_m(m).with_epsilon(0.0).with_max_iterations(iters)
    if owl:
        b = b.with_orthantwise(*owl)
    b.minimize(x, O.logistic() if owl else O.quadratic(), lambda p: rows.append((p["fx"], p["gnorm"], p["step"])) and False)
    return rows, x


def run_device(ctx, n, m, iters, owl=None):
    rows, x = [], np.zeros(n)
    b = R.lbfgs().with_m(m).with_epsilon(0.0).with_max_iterations(iters)
    if owl:
        b = b.with_orthantwise(*owl)
    b.minimize(x, objectives.Logistic() if owl else objectives.Quadratic(), lambda p: rows.append((p.fx, p.gnorm, p.step)) and False,
               ctx=ctx)
    return rows, x


def close(rows_o, rows_g, xo, xg, tol=1e-10):
    assert len(rows_o) == len(rows_g)
    for ro, rg in zip(rows_o, rows_g):
        for a, b in zip(ro, rg):
            assert abs(a - b) <= tol * max(abs(a), 1e-300), (ro, rg)
    assert np.max(np.abs(xo - xg)) <= tol * max(np.max(np.abs(xo)), 1e-300)


@pytest.mark.parametrize("owl", [None, (0.5, 1000, 250_000)], ids=["lbfgs", "owlqn"])
def test_a_resident_launch_that_loses_a_workgroup_is_rerun_per_step(owl, monkeypatch, capfd):
    """LBFGS_HIP_RESIDENT_FAULT=1: the first resident launch's last workgroup never takes part (as if it had never been
    given a CU).  The others give up after the hand-off timeout, the error word reaches the host with the results, and
    the read that finds it re-runs the recursion with a kernel per step: same trajectory as the oracle, one warning,
    no resident launch afterwards."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    n, m, iters = 300_007, 5, 12
    monkeypatch.setenv("LBFGS_HIP_RESIDENT_FAULT", "1")
    monkeypatch.setenv("LBFGS_HIP_HANDOFF_TIMEOUT_MS", "100")
    rows_o, xo = run_oracle(n, m, iters, owl)
    with R.Context(n) as ctx:
        rows_g, xg = run_device(ctx, n, m, iters, owl)
        resident = ctx.resident_two_loops()
    close(rows_o, rows_g, xo, xg)
    assert resident == 1, resident          # the faulty one; every later two-loop took the kernel-per-step path
    err = capfd.readouterr().err
    assert err.count("timed out waiting for a workgroup") == 1, err


def test_a_mis_detected_chip_costs_milliseconds_not_the_full_timeout(monkeypatch, capfd):
    """Until a resident launch of a context has been SEEN to complete, its hand-offs wait 50 ms at most
    (LBFGS_HIP_RESIDENT_FIRST_TIMEOUT_MS): a device that cannot hold the grid resident (partitioned, shared, CU-masked in a way
    the probes miss) falls back to the kernel-per-step path after milliseconds.  Here: the default 10 s hand-off timeout is left
    alone, the first launch loses a workgroup -- and the whole run, fallback included, must take a fraction of that."""
    import time

    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    n, m, iters = 300_007, 5, 12
    monkeypatch.setenv("LBFGS_HIP_RESIDENT_FAULT", "1")
    monkeypatch.delenv("LBFGS_HIP_HANDOFF_TIMEOUT_MS", raising=False)
    rows_o, xo = run_oracle(n, m, iters)
    with R.Context(n) as ctx:
        t0 = time.perf_counter()
        rows_g, xg = run_device(ctx, n, m, iters)
        wall = time.perf_counter() - t0
        resident, info = ctx.resident_two_loops(), ctx.comm_info()
    close(rows_o, rows_g, xo, xg)
    assert resident == 1 and info["resident_fallbacks"] == 1
    assert wall < 3.0, wall  # (10 s with the configured timeout)
    assert capfd.readouterr().err.count("timed out waiting for a workgroup") == 1


def test_the_configured_timeout_applies_once_a_resident_launch_has_completed(monkeypatch):
    """... and once a resident launch HAS completed, the context is trusted with the configured timeout -- half of it: a resident
    hand-off gives up before a launch-per-step reduction of another context on the same GPU would (see
    test_two_contexts_on_two_streams_of_one_gpu).  A later launch that loses a workgroup waits that long (here 200 ms of the
    configured 400: longer than the first-launch bound, short enough for a test)."""
    import time

    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    n, m = 300_007, 4
    monkeypatch.setenv("LBFGS_HIP_HANDOFF_TIMEOUT_MS", "400")
    monkeypatch.setenv("LBFGS_HIP_RESIDENT_FAULT", "2")  # (the SECOND resident launch of the context loses a workgroup)
    with R.Context(n) as ctx:
        hist, g, d = _history_with_pairs(ctx, m)
        hist.two_loop(d, g, m, m - 1, 7, 8, 12)
        ref_dn = ctx.scalars(12, 2).copy()      # (this read proves the launch)
        assert ctx.resident_two_loops() == 1
        t0 = time.perf_counter()
        hist.two_loop(d, g, m, m - 1, 7, 8, 12)
        dn = ctx.scalars(12, 2).copy()
        wall = time.perf_counter() - t0
        assert 0.18 < wall < 3.0, wall
        assert ctx.comm_info()["resident_fallbacks"] == 1
        np.testing.assert_allclose(dn, ref_dn, rtol=1e-12)
        hist.free(); g.free(); d.free()


def _history_with_pairs(ctx, m):
    """m corrections with y = A s of the hashed quadratic (ys > 0), and a gradient: inputs of a two-loop, all on the device"""
    from rust_lbfgs_amd import hotpath as H
    from rust_lbfgs_amd.math import DeviceVec

    hist = H.History(ctx, m)
    g, d, tmp = DeviceVec(ctx), DeviceVec(ctx), DeviceVec(ctx)
    q = objectives.Quadratic()
    for j in range(m):
        tmp.fill(0.25 + 0.1 * j)
        H.objective_eval(q, tmp, hist.s(j), 0)
        H.objective_eval(q, hist.s(j), hist.y(j), 0)
        hist.y(j).vecadd(hist.s(j), 2.0)
    ys = [hist.y(j).vecdot(hist.s(j)) for j in range(m)]
    hist.set_scalars(ys=np.array(ys), alpha=np.zeros(m))
    tmp.fill(-0.3)
    H.objective_eval(objectives.Logistic(), tmp, g, 0)
    ctx.set_scalars(7, [ys[m - 1], hist.y(m - 1).vecdot(hist.y(m - 1))])
    tmp.free()
    return hist, g, d


@pytest.mark.parametrize("first", ["download", "history_scalars", "sync"])
def test_every_synchronising_entry_point_recovers_from_a_timed_out_resident_launch(first, monkeypatch, capfd):
    """Round-3 advice: the recovery used to live in lbfgs_hip_scalars_read only, so a caller of the C-ABI that downloaded d (or
    read alpha, or just synchronised) FIRST got the aborted kernel's output with rc = OK.  Now every entry point that hands
    results of the stream to the host looks at the device error word: the direction downloaded right after a resident launch
    that lost a workgroup is the kernel-per-step path's, to the last bit of what that path gives on the same inputs."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    n, m = 300_007, 5
    monkeypatch.setenv("LBFGS_HIP_RESIDENT", "0")
    with R.Context(n) as ctx:
        hist, g, d = _history_with_pairs(ctx, m)
        hist.two_loop(d, g, m, m - 1, 7, 8, 12)
        ref_d, ref_dn, ref_alpha = d.to_numpy(), ctx.scalars(12, 2).copy(), hist.scalars()[1].copy()
        hist.free(); g.free(); d.free()
    monkeypatch.setenv("LBFGS_HIP_RESIDENT", "1")
    monkeypatch.setenv("LBFGS_HIP_RESIDENT_FAULT", "1")
    with R.Context(n) as ctx:
        hist, g, d = _history_with_pairs(ctx, m)
        hist.two_loop(d, g, m, m - 1, 7, 8, 12)
        assert ctx.resident_two_loops() == 1
        if first == "download":
            got_d = d.to_numpy()
        elif first == "history_scalars":
            got_alpha = hist.scalars()[1].copy()
            np.testing.assert_array_equal(got_alpha, ref_alpha)
            got_d = d.to_numpy()
        else:
            ctx.sync()
            got_d = d.to_numpy()
        np.testing.assert_array_equal(got_d, ref_d)
        np.testing.assert_array_equal(ctx.scalars(12, 2), ref_dn)
        assert ctx.comm_info()["resident_fallbacks"] == 1
        hist.free(); g.free(); d.free()
    assert capfd.readouterr().err.count("timed out waiting for a workgroup") == 1


@pytest.mark.parametrize("change", ["upload_g", "write_gamma", "write_ys"])
def test_inputs_changed_after_a_resident_launch_are_never_silently_re_used(change, monkeypatch):
    """... and the recovery re-runs the recursion on its INPUTS: a call that changes one of them after the launch (an upload
    into g, a write of the gamma slots, of ys) withdraws the permission to re-run -- the timed-out launch then surfaces as the
    error it is instead of a recomputation from other inputs."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    n, m = 300_007, 5
    monkeypatch.setenv("LBFGS_HIP_RESIDENT_FAULT", "1")
    ctx = R.Context(n)
    try:
        hist, g, d = _history_with_pairs(ctx, m)
        hist.two_loop(d, g, m, m - 1, 7, 8, 12)
        if change == "upload_g":
            g.upload(np.ones(n))
        elif change == "write_gamma":
            ctx.set_scalars(7, [1.0, 2.0])
        else:
            hist.set_scalars(ys=np.ones(m))
        with pytest.raises(R.LbfgsError, match="timed out waiting for a workgroup"):
            ctx.scalars(12, 2)
        assert ctx.comm_info is not None
    finally:
        ctx.close()


def test_two_contexts_on_two_streams_of_one_gpu(monkeypatch):
    """Two independent optimisations in ONE process, each with its own context and stream, both eligible for the
    chip-wide kernel, driven from two threads at the same time.  Whatever the dispatcher does with two kernels that each
    want every CU -- one after the other, or a share each (then both time out once and fall back) -- both runs must
    finish with the oracle's trajectory.  That includes the case in which one context's resident kernel and the OTHER
    context's launch-per-step reduction block each other (the resident kernel needs the CUs empty on which the other kernel's
    last workgroup waits for workgroups that are bound to XCDs the resident kernel fills): the resident kernel's hand-offs
    wait half the configured timeout, so it is always the one that gives up -- and it is the one that can be re-run."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    n, m, iters = 2_000_003, 6, 25
    monkeypatch.setenv("LBFGS_HIP_HANDOFF_TIMEOUT_MS", "300")
    rows_o, xo = run_oracle(n, m, iters)
    out, errs = [None, None], []

    def work(i, ctx):
        try:
            out[i] = run_device(ctx, n, m, iters)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    with R.Context(n) as c0, R.Context(n) as c1:
        ts = [threading.Thread(target=work, args=(i, c)) for i, c in enumerate((c0, c1))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in ts)
    assert not errs, errs
    for rows_g, xg in out:
        close(rows_o, rows_g, xo, xg)


# ------------------------------------------------------------------------------------------------------------
# environment knobs: every setting that selects another code path must give the oracle's trajectory
# ------------------------------------------------------------------------------------------------------------
_ORACLE_RUNS = {}
KNOBS = [
    {},                                                   # defaults
    {"LBFGS_HIP_GRAN_CACHED": "1"},                       # granules in plain hipMalloc memory
    {"LBFGS_HIP_RESIDENT_PLAIN_MB": "0"},                 # hybrid: no cache slice, everything streamed
    {"LBFGS_HIP_RESIDENT_PLAIN_MB": "1"},                 # hybrid: a slice of 32 rounds, the rest `nt` (both halves of res_hbm_rounds)
    {"LBFGS_HIP_RESIDENT_HYBRID": "0"},                   # shards beyond the chip: kernel per step
    {"LBFGS_HIP_RESIDENT": "0"},                          # never the persistent kernel
    {"LBFGS_HIP_RESIDENT": "0", "LBFGS_HIP_DEFER_SUMS": "0"},
    {"LBFGS_HIP_HANDOFF": "ticket"},                      # arrival-counter reductions (the persistent kernel is not eligible)
    {"LBFGS_HIP_NO_MIRROR": "1"},                         # scalar reads by copy
    {"LBFGS_HIP_RESIDENT_NT_MB": "100000"},               # the persistent kernel without `nt` hints
    {"LBFGS_HIP_RESIDENT_TOUCH": "0"},                    # the waiting workgroups touch (next to) nothing ahead
    {"LBFGS_HIP_RESIDENT_TOUCH": "16"},                   # ... the deepest touch, whatever the shard size
    {"LBFGS_HIP_GRID": "64"},                             # a launch grid for the streaming kernels (takes the persistent kernel out)
    {"LBFGS_HIP_NT_THRESHOLD_MB": "1", "LBFGS_HIP_NT_STORE_THRESHOLD_MB": "1", "LBFGS_HIP_RESIDENT": "0"},  # `nt` everywhere
]


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join(f"{a[10:]}={b}" for a, b in k.items()) or "defaults")
@pytest.mark.parametrize("shape", ["on_chip", "hybrid_streaming"])
def test_every_knob_setting_follows_the_oracle(knobs, shape, monkeypatch):
    """on_chip: n = 300 007 on the default grid (the whole running vector in registers + LDS).  hybrid_streaming: n = 3.3e6
    on 8 workgroups -- 26 MB vectors carry the `nt` hints, and 8 workgroups hold 3.9e5 elements, so 88 % of q lives in HBM:
    the form n = 1e8 takes on a whole GPU, cache slice and alternating sweep included."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    if shape == "on_chip":
        n, m, iters = 300_007, 6, 12
    else:
        n, m, iters = 3_300_001, 5, 9
        monkeypatch.setenv("LBFGS_HIP_RESIDENT_GRID", "8")
    if (n, m, iters) not in _ORACLE_RUNS:
        _ORACLE_RUNS[(n, m, iters)] = run_oracle(n, m, iters)
    rows_o, xo = _ORACLE_RUNS[(n, m, iters)]
    with R.Context(n) as ctx:
        rows_g, xg = run_device(ctx, n, m, iters)
        resident, on_chip = ctx.resident_two_loops(), ctx.resident_elements()
    close(rows_o, rows_g, xo, xg)
    eligible = knobs.get("LBFGS_HIP_RESIDENT") != "0" and knobs.get("LBFGS_HIP_HANDOFF") != "ticket" and "LBFGS_HIP_GRID" not in knobs
    if shape == "hybrid_streaming" and knobs.get("LBFGS_HIP_RESIDENT_HYBRID") == "0":
        eligible = False
    assert (resident > 0) == eligible, (resident, knobs)
    if eligible and shape == "hybrid_streaming":
        assert on_chip == 2 * 96 * 256 * 8
