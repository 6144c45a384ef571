"""N > 1 path on the CPU: two gloo ranks, contiguous shards, scalar all-reduce through the callback
communicator, on the CPU test double of the C-ABI (tests/support).  Checks that the SHARDED run
reproduces the single-rank oracle trajectory (to summation-order noise: each rank sums its shard
sequentially, then the two partials are added), for L-BFGS and for OWL-QN with an [start, end)
range that straddles the shard boundary, and that a sharded host closure's partial f is reduced."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["LBFGS_ROOT"])
import torch.distributed as dist
import rust_lbfgs_amd as R
from rust_lbfgs_amd import _ffi, objectives
from rust_lbfgs_amd import dist as D
if os.environ.get("LBFGS_WORKER_PRODUCT") != "1":   # CPU suite: the test double; GPU suite: the HIP library
    from tests.support import mock
    mock.install()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
case = json.loads(os.environ["LBFGS_CASE"])
n = case["n"]
bounds = json.loads(os.environ.get("LBFGS_TEST_BOUNDS", "null"))     # uneven shards: [0, b1, ..., n]
if os.environ.get("LBFGS_TEST_RESIDENT_GRIDS"):                      # workgroups of the resident kernel, per rank
    os.environ["LBFGS_HIP_RESIDENT_GRID"] = os.environ["LBFGS_TEST_RESIDENT_GRIDS"].split(",")[rank]
# (LBFGS_TEST_DEVICE_PER_RANK=1: one GPU per rank, as on a real node -- tests/test_gpu_multi_device.py; otherwise device 0)
dev = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("LBFGS_TEST_DEVICE_PER_RANK") == "1" else 0
ctx = D.sharded_context(n, device=dev, kind=os.environ.get("LBFGS_COMM_KIND", "callback"),
                        exclusive_device=os.environ.get("LBFGS_TEST_EXCLUSIVE_DEVICE") == "1", bounds=bounds)
lo, hi = (bounds[rank], bounds[rank + 1]) if bounds else D.shard_range(n, rank, world)
assert ctx.n_local == hi - lo and ctx.shard.offset == lo
b = R.lbfgs().with_m(case["m"]).with_max_iterations(case["iters"]).with_epsilon(0.0)
if case.get("owl"):
    b = b.with_orthantwise(*case["owl"])
if case["objective"] == "fuzz":
    from tests import fuzz_common as F
    fc = case["fuzz"]
    b = F.configure(R.lbfgs(), fc)
    if fc.get("vector_free") and fc["m"] <= 10:
        b = b.with_vector_free(True)
    ev = {"quadratic": objectives.Quadratic, "logistic": objectives.Logistic, "rosenbrock": objectives.Rosenbrock}[fc["kind"]]()
elif case["objective"] == "quadratic":
    ev = objectives.Quadratic()
elif case["objective"] == "logistic":
    ev = objectives.Logistic()
else:
    # a separable HOST closure evaluated on this rank's shard only: returns the shard's partial f
    def ev(x, g):
        idx = np.arange(lo, hi, dtype=np.float64)
        a = 1.0 + (idx % 7.0)
        g[:] = a * (x - 0.5)
        return float(np.sum(0.5 * a * (x - 0.5) ** 2))
rows = []
x = np.zeros(ctx.n_local)
err = 0
errmsg = ""
if case["objective"] == "fuzz":
    x = F.x0_of(fc)[lo:hi].copy()
try:
    st = b.build(x, ev, ctx=ctx)
    try:
        while not st.is_converged():
            p = st.propagate()
            rows.append([p.niter, p.neval, p.ncall, p.fx, p.xnorm, p.gnorm, p.step])
    except R.LbfgsError as e:
        err = e.code
        errmsg = str(e)
    xs = st.download("x")
    st.close()
except R.LbfgsError as e:
    err = e.code
    errmsg = str(e)
    xs = x
placement = getattr(ctx, "p2p_placement", None)
ctx_resident = ctx.resident_two_loops()
ctx_resident_elements = ctx.resident_elements()
nred, _ = ctx.prof_read(_ffi.K_COMM)   # the test double counts its all-reduces here
if os.environ.get("LBFGS_WORKER_PRODUCT") == "1":
    nred = 1
ctx.close()
out = dict(rank=rank, lo=lo, hi=hi, rows=rows, x=xs.tolist(), allreduces=nred, err=err,
           resident=ctx_resident, resident_elements=ctx_resident_elements, placement=placement, errmsg=errmsg)
json.dump(out, open(os.path.join(os.environ["LBFGS_OUT"], f"rank{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(case, world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LBFGS_ROOT=ROOT, LBFGS_CASE=json.dumps(case), LBFGS_OUT=str(tmp_path),
               OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world)]


def oracle_rows(case):
    from oracle import oracle as O

    b = O.lbfgs().with_m(case["m"]).with_max_iterations(case["iters"]).with_epsilon(0.0)
    if case.get("owl"):
        b = b.with_orthantwise(*case["owl"])
    n = case["n"]
    if case["objective"] == "quadratic":
        ev = O.quadratic()
    elif case["objective"] == "logistic":
        ev = O.logistic()
    else:
        def ev(x, g):
            idx = np.arange(0, n, dtype=np.float64)
            a = 1.0 + (idx % 7.0)
            g[:] = a * (x - 0.5)
            return float(np.sum(0.5 * a * (x - 0.5) ** 2))
    rows = []
    x = np.zeros(n)
    b.minimize(x, ev, lambda p: rows.append([p["niter"], p["neval"], p["ncall"], p["fx"], p["xnorm"], p["gnorm"],
                                            p["step"]]) and False)
    return rows, x


@pytest.mark.parametrize("case", [
    dict(name="quadratic", n=5000, m=5, iters=25, objective="quadratic"),
    dict(name="owlqn_straddle", n=3001, m=6, iters=20, objective="logistic", owl=[0.5, 700, 2900]),
    dict(name="host_closure", n=1000, m=4, iters=12, objective="closure"),
], ids=lambda c: c["name"])
def test_two_ranks_match_single_rank_oracle(case, tmp_path):
    outs = run_world(case, 2, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    # shards tile [0, n) contiguously
    assert outs[0]["lo"] == 0 and outs[0]["hi"] == outs[1]["lo"] and outs[1]["hi"] == case["n"]
    # every rank saw the same (global) scalars
    assert outs[0]["rows"] == outs[1]["rows"]
    assert outs[0]["allreduces"] == outs[1]["allreduces"] > 0
    rows = outs[0]["rows"]
    assert len(rows) == len(ref_rows)
    for got, ref in zip(rows, ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6), (got, ref)
    x = np.concatenate([np.array(o["x"]) for o in outs])
    assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)


def test_shard_range_properties():
    import rust_lbfgs_amd as R  # noqa: F401
    from rust_lbfgs_amd.dist import shard_range

    for n in (0, 1, 255, 256, 257, 1000, 10**8, 10**8 + 1):
        for world in (1, 2, 3, 4, 8):
            pieces = [shard_range(n, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            for (a, b), (c, d) in zip(pieces, pieces[1:]):
                assert b == c and a <= b
            for lo, hi in pieces:
                if hi > lo:  # non-empty shards start on a 256-element (2 KiB) boundary
                    assert lo % 256 == 0 and (hi % 256 == 0 or hi == n)
    assert shard_range(10**8, 7, 8) == (87501568, 100000000)


@pytest.mark.parametrize("seed", [3, 7, 12, 21, 33])
def test_random_configurations_two_ranks(seed, tmp_path):
    """The seeded random sweep (tests/fuzz_common.py) SHARDED over two gloo ranks against the single-rank oracle:
    same error code and discrete decisions, values within the run's calibrated tolerance."""
    compare_sharded_fuzz(seed, tmp_path, vector_free=False)


def compare_sharded_fuzz(seed, tmp_path, vector_free):
    from tests import fuzz_common as F

    c = F.make_case(seed)
    c["n"] = max(c["n"], 600)  # two non-empty shards (boundaries are multiples of 256)
    if c["kind"] == "rosenbrock":
        c["n"] -= c["n"] % 2
    c["vector_free"] = vector_free
    ro, xo, eo = F.run_oracle(c, 0)
    floors, _, all_stable = F.order_sensitivity(c, ro, eo)
    outs = run_world(dict(name=f"fuzz{seed}", n=c["n"], m=c["m"], iters=c["iters"], objective="fuzz", fuzz=c), 2, tmp_path)
    assert outs[0]["rows"] == outs[1]["rows"] and outs[0]["err"] == outs[1]["err"]
    F.compare_with_oracle(c, ro, eo, outs[0]["rows"], outs[0]["err"], floors, all_stable, slack=5.0 if vector_free else 1.0)


def test_three_ranks_uneven_shards(tmp_path):
    """world = 3: the last shard is shorter (and 256-aligned boundaries leave it ragged)."""
    case = dict(name="quadratic3", n=1000, m=4, iters=15, objective="quadratic")
    outs = run_world(case, 3, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    assert [o["hi"] - o["lo"] for o in outs] == [512, 488, 0]  # ceil(1000/3) -> 334 -> 512-aligned shards
    assert outs[0]["rows"] == outs[1]["rows"] == outs[2]["rows"]
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6)
    x = np.concatenate([np.array(o["x"]) for o in outs])
    assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)


def test_explicit_shard_bounds(tmp_path, monkeypatch):
    """`sharded_context(bounds=...)`: shards need not be even -- a large one, a sliver and an EMPTY one; OWL-QN's range straddles
    the first boundary.  Same trajectory as the single-rank oracle; invalid bounds are refused."""
    monkeypatch.setenv("LBFGS_TEST_BOUNDS", json.dumps([0, 2817, 3001, 3001]))
    case = dict(name="owlqn_bounds", n=3001, m=6, iters=20, objective="logistic", owl=[0.5, 700, 2900])
    outs = run_world(case, 3, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    assert [(o["lo"], o["hi"]) for o in outs] == [(0, 2817), (2817, 3001), (3001, 3001)]
    assert outs[0]["rows"] == outs[1]["rows"] == outs[2]["rows"] and all(o["err"] == 0 for o in outs)
    assert len(outs[0]["rows"]) == len(ref_rows)
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6), (got, ref)
    x = np.concatenate([np.array(o["x"]) for o in outs])
    assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)
    import rust_lbfgs_amd as R  # noqa: F401
    from rust_lbfgs_amd import dist as D

    class FakeDist:  # (bounds are validated before anything touches a process group)
        pass

    import torch.distributed as tdist

    monkeypatch.setattr(tdist, "get_rank", lambda g=None: 0)
    monkeypatch.setattr(tdist, "get_world_size", lambda g=None: 2)
    for bad in ([0, 10], [1, 5, 10], [0, 7, 5, 10][:3], [0, 11, 10]):
        with pytest.raises(ValueError):
            D.sharded_context(10, kind="callback", bounds=bad)


def test_eight_ranks_as_in_the_metric(tmp_path):
    """world = 8, the rank count BASELINE.json's metric names: eight contiguous 256-aligned shards (the last one
    short), OWL-QN range straddling several shard boundaries, every rank holding the same global scalars."""
    case = dict(name="owlqn8", n=20_011, m=6, iters=12, objective="logistic", owl=[0.5, 3000, 17_500])
    outs = run_world(case, 8, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    sizes = [o["hi"] - o["lo"] for o in outs]
    assert sizes == [2560] * 7 + [20_011 - 7 * 2560] and outs[0]["lo"] == 0 and outs[-1]["hi"] == case["n"]
    for o in outs[1:]:
        assert o["rows"] == outs[0]["rows"]
    assert len(outs[0]["rows"]) == len(ref_rows)
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6)
    x = np.concatenate([np.array(o["x"]) for o in outs])
    assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)


# ------------------------------------------------------------------------------------------------------------
# bench.py's N > 1 supervisor: rank processes run on the test double (tests/support/bench_on_mock.py)
# ------------------------------------------------------------------------------------------------------------
def _run_bench(extra_args, launcher, legs=None, timeout=300):
    # (these tests are about the supervisor and a double WITHOUT communicators of its own: its shared-memory stand-ins of P2P and
    # RCCL -- round 6 -- stay off here; tests that want them ask for them)
    env = dict(os.environ, LBFGS_BENCH_WORKER=os.path.join(ROOT, "tests", "support", "bench_on_mock.py"),
               OMP_NUM_THREADS="1", LBFGS_MOCK_NO_P2P="1")
    env.pop("LBFGS_MOCK_RCCL", None)
    if legs:
        env["LBFGS_BENCH_LEGS"] = legs
    args = ["--gpus", "2", "--steps", "4", "--warmup", "12", "--dim", "3000", "--hist", "5", "--repeats", "2",
            "--no-vector-free"] + extra_args
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("launcher", [False, True], ids=["plain_python", "torch_distributed_run"])
def test_bench_supervisor_two_ranks(launcher):
    """`python bench.py --gpus 2` and the torch.distributed.run form both end in exactly ONE JSON line."""
    p = _run_bench(["--comm", "callback"], launcher)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 12 and j["value"] > 0 and j["scaling"] == "strong"
    assert j["config"]["legs"]["callback"]["status"] == "ok" and j["config"]["allreduce"] == "callback"
    assert j["config"]["repeats"] == 2 and len(j["config"]["repeats_iters_per_sec"]) == 2
    assert j["config"]["n_local_rank0"] == 1536  # shard_range(3000, 0, 2)


@pytest.mark.parametrize("launcher", [False, True], ids=["plain_python", "torch_distributed_run"])
def test_bench_supervisor_survives_a_hung_and_a_failed_leg(launcher):
    """A leg that never returns is killed at its timeout, a communicator that does not exist (the test double has no RCCL)
    fails its PROBE and is never measured, and the run still prints the line of the leg that worked and exits 0."""
    p = _run_bench(["--leg-timeout", "15"], launcher, legs="hang,rccl,callback")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    cfg = json.loads(lines[0])["config"]
    legs, probes = cfg["legs"], cfg["probes"]
    assert probes["hang"]["status"] == "ok" and probes["rccl"]["status"] != "ok" and probes["callback"]["status"] == "ok"
    assert "timed out" in legs["hang"]["status"] and "rccl" not in legs and legs["callback"]["status"] == "ok"


def test_bench_supervisor_default_leg_order():
    """No --comm: the communicators are probed in the order p2p, p2p-host, rccl (+ rccl-per-step when rccl failed); the test double has none of them, so no leg
    is measured and the host-staged callback runs as the last resort -- only because nothing else produced a result."""
    p = _run_bench(["--probe-timeout", "60"], False)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    cfg = json.loads(lines[0])["config"]
    # (a failed rccl probe -- the gated exchange is that leg's default form -- makes the supervisor probe RCCL's plain form too)
    assert list(cfg["probes"]) == ["p2p", "p2p-host", "rccl", "rccl-per-step"]
    assert all(v["status"] != "ok" for v in cfg["probes"].values())
    assert list(cfg["legs"]) == ["callback"] and cfg["legs"]["callback"]["status"] == "ok"


@pytest.mark.parametrize("launcher", [True], ids=["torch_distributed_run"])   # (the driver's launch form for N > 1)
def test_bench_supervisor_two_hung_legs_stay_inside_the_budget(launcher):
    """Two legs in a row that never return: each is killed at ITS SHARE of what is left of --total-budget (not at a fixed
    per-leg timeout), so the leg behind them still gets its turn and the whole run ends inside the budget with a line."""
    import time

    t0 = time.monotonic()
    p = _run_bench(["--total-budget", "80", "--leg-timeout", "60"], launcher, legs="hang,hang2,callback", timeout=200)
    wall = time.monotonic() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    cfg = json.loads(lines[0])["config"]
    legs = cfg["legs"]
    assert "timed out" in legs["hang"]["status"] and "timed out" in legs["hang2"]["status"] and legs["callback"]["status"] == "ok"
    # the shares: a third, then half of what was left (never the 60 s the option would allow)
    assert legs["hang"]["timeout_s"] <= 27.0 and legs["hang2"]["timeout_s"] <= 32.0, legs
    assert cfg["budget"]["used_s"] <= 80.0 and wall < 105.0, (cfg["budget"], wall)


def test_bench_supervisor_failed_or_hung_probe_skips_its_leg():
    """Only communicators whose probe passed are measured: a probe that fails and a probe that never returns (killed at the
    probe timeout) both keep their legs out of phase 2."""
    p = _run_bench(["--probe-timeout", "6"], False, legs="failprobe,hangprobe,callback")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    cfg = json.loads(lines[0])["config"]
    assert cfg["probes"]["failprobe"]["status"] != "ok" and "timed out" in cfg["probes"]["hangprobe"]["status"]
    assert cfg["probes"]["callback"]["status"] == "ok"
    assert list(cfg["legs"]) == ["callback"] and cfg["legs"]["callback"]["status"] == "ok"


def test_bench_supervisor_sigterm_prints_the_best_line_so_far():
    """The driver's clock runs out in the middle of a leg: SIGTERM ends the running job and the run prints the line of the
    legs that HAVE finished (exit code 0), instead of dying with nothing."""
    import signal
    import threading
    import time

    env = dict(os.environ, LBFGS_BENCH_WORKER=os.path.join(ROOT, "tests", "support", "bench_on_mock.py"),
               OMP_NUM_THREADS="1", LBFGS_BENCH_LEGS="callback,hang")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "12", "--dim", "3000",
           "--hist", "5", "--repeats", "2", "--no-vector-free", "--total-budget", "400", "--leg-timeout", "300"]
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    seen, err_lines = threading.Event(), []

    def watch():
        for ln in p.stderr:
            err_lines.append(ln)
            if "[bench] leg callback" in ln:
                seen.set()

    th = threading.Thread(target=watch, daemon=True)
    th.start()
    assert seen.wait(timeout=150), "".join(err_lines)[-2000:]
    time.sleep(2.0)  # the hung leg is running now
    p.send_signal(signal.SIGTERM)
    out = p.stdout.read()
    rc = p.wait(timeout=60)
    th.join(timeout=10)
    assert rc == 0, "".join(err_lines)[-2000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    j = json.loads(lines[0])
    assert j["value"] > 0 and j["config"]["interrupted_by_signal"] == signal.SIGTERM
    assert j["config"]["legs"]["callback"]["status"] == "ok" and "hang" not in j["config"]["legs"]
    # nothing of the hung job is left behind: its whole process group is gone
    import re

    m = re.search(r"running job's process group: (\d+)", "".join(err_lines))
    assert m, "".join(err_lines)[-1000:]
    time.sleep(0.5)
    with pytest.raises(ProcessLookupError):
        os.killpg(int(m.group(1)), 0)


def test_eight_ranks_as_processes_times_threads_tool_on_the_test_double(tmp_path):
    """tools/eight_ranks_one_gpu.py -- world 8 as 4 processes x 2 host threads behind a rendezvous made of files, bench.py's own
    make_context + measure per rank -- dry-run on the CPU test double (callback communicator; the GPU suite runs it with the P2P
    exchange inside the persistent kernel): the duck-typed process group of rust_lbfgs_amd.dist, eight shards, trajectories
    equal to the oracle's, one bench line."""
    out = os.path.join(ROOT, "gpurun_out", "bench_eight_ranks_sharing_one_gpu.json")
    keep = out + ".kept_by_the_cpu_test"
    if os.path.exists(out):
        os.replace(out, keep)
    try:
        env = dict(os.environ, LBFGS_TEST_BACKEND="mock", LBFGS_EIGHT_SMALL="1", OMP_NUM_THREADS="1")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eight_ranks_one_gpu.py"), "--dim", "40003", "--hist", "5", "--steps",
                            "4", "--repeats", "2", "--legs", "callback"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        rec = json.load(open(out))
    finally:
        if os.path.exists(keep):
            os.replace(keep, out)
        elif os.path.exists(out):
            os.unlink(out)
    assert rec["ok"] and rec["exit_codes"] == [0, 0, 0, 0]
    assert len(rec["result"]["trajectories"]) == 3 and all(t["ok"] and t["identical_rows_on_every_rank"] for t in rec["result"]["trajectories"])
    line = rec["result"]["legs"]["callback"]["line"]
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["config"]["comm_info"]["world"] == 8
    assert line["config"]["comm_info"]["exchanges_per_two_loop"] == pytest.approx(12.0)


# ---------------------------------------------------------------------------------------------
# Round 6: first contact with several ranks, blind.  The test double now has stand-ins for the P2P and the RCCL communicator over
# POSIX shared memory (tests/support/mock_lbfgs_hip.cpp "bus"): real rank PROCESSES run dist.py's communicator code -- the handle
# exchange, the collective fall-back from device- to host-placed mailboxes, the sealing -- and the product's OWN collective
# skeleton of the gated RCCL exchange's preparation (rust-lbfgs_amd/csrc/ext_protocol.h) with failures injected on ONE rank,
# which no single GPU can host.
# ---------------------------------------------------------------------------------------------
def run_world_env(case, world, tmp_path, extra_env, ok_codes=(0,)):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LBFGS_ROOT=ROOT, LBFGS_CASE=json.dumps(case), LBFGS_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode in ok_codes, r.stderr[-3000:]
    return r, [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world) if (tmp_path / f"rank{k}.json").exists()]


def _same_as_oracle(case, outs):
    ref_rows, ref_x = oracle_rows(case)
    for o in outs[1:]:
        assert o["rows"] == outs[0]["rows"]
    rows = outs[0]["rows"]
    assert len(rows) == len(ref_rows)
    for got, ref in zip(rows, ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6), (got, ref)
    x = np.concatenate([np.array(o["x"])[: o["hi"] - o["lo"]] for o in outs])
    assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)


@pytest.mark.parametrize("kind,world,env,placement", [
    ("p2p", 3, {}, "device"),
    ("p2p-host", 2, {}, "host"),
    # every rank sees only its OWN device (a launcher that masks devices per rank): a peer's device-placed mailbox cannot be
    # mapped -- every rank learns that some rank failed and all of them retry with host-placed mailboxes, collectively
    ("p2p", 3, {"LBFGS_MOCK_NO_DEVICE_IPC": "1"}, "host"),
    ("rccl", 3, {"LBFGS_MOCK_RCCL": "1"}, None),
], ids=["p2p_device", "p2p_host", "p2p_masked_devices_fall_back_to_host", "rccl"])
def test_bus_communicators_of_the_test_double_run_whole_optimisations(kind, world, env, placement, tmp_path):
    case = dict(name="quadratic", n=5000, m=5, iters=18, objective="quadratic")
    r, outs = run_world_env(case, world, tmp_path, dict(env, LBFGS_COMM_KIND=kind, LBFGS_TEST_EXCLUSIVE_DEVICE="1"))
    assert len(outs) == world and all(o["err"] == 0 for o in outs), [o["errmsg"] for o in outs]
    assert all(o["placement"] == placement for o in outs)
    if env.get("LBFGS_MOCK_NO_DEVICE_IPC"):
        assert "retrying with host-placed mailboxes" in r.stderr
    _same_as_oracle(case, outs)


def test_p2p_device_placement_is_strict_when_asked_for(tmp_path):
    """bench.py's "p2p" leg means mailboxes in DEVICE memory (kind "p2p-device": no silent retry -- host placement is its own
    leg): with masked devices every rank must fail together, with the IPC error, and nobody may hang."""
    case = dict(name="quadratic", n=5000, m=5, iters=5, objective="quadratic")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.replace('ctx = D.sharded_context(', 'import time\ntry:\n    ctx = D.sharded_context(', 1).replace(
        'exclusive_device=os.environ.get("LBFGS_TEST_EXCLUSIVE_DEVICE") == "1", bounds=bounds)',
        'exclusive_device=os.environ.get("LBFGS_TEST_EXCLUSIVE_DEVICE") == "1", bounds=bounds)\n'
        'except R.LbfgsError as e:\n'
        '    json.dump(dict(rank=rank, refused=str(e)), open(os.path.join(os.environ["LBFGS_OUT"], f"rank{rank}.json"), "w"))\n'
        '    dist.barrier(); dist.destroy_process_group(); sys.exit(0)', 1))
    env = dict(os.environ, LBFGS_ROOT=ROOT, LBFGS_CASE=json.dumps(case), LBFGS_OUT=str(tmp_path), OMP_NUM_THREADS="1",
               LBFGS_COMM_KIND="p2p-device", LBFGS_MOCK_NO_DEVICE_IPC="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    outs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(3)]
    assert all("refused" in o for o in outs)
    assert any("hipIpcOpenMemHandle" in o["refused"] for o in outs) and all("P2P" in o["refused"] or "mailbox" in o["refused"] for o in outs)


GATED_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["LBFGS_ROOT"])
import numpy as np
import torch.distributed as dist
import rust_lbfgs_amd as R
from rust_lbfgs_amd import _ffi, objectives
from rust_lbfgs_amd import dist as D
from tests.support import mock
mock.install()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = int(os.environ["LBFGS_TEST_N"])
out = dict(rank=rank)
ctx = None
try:
    ctx = D.sharded_context(n, kind="rccl", exclusive_device=True)
    out["resident_form"] = ctx.prof_read(_ffi.K_TWOLOOP_RESIDENT)[0] > 0   # (the double reports the form the product would take)
    x = np.zeros(ctx.n_local)
    rows = []
    R.lbfgs().with_m(4).with_epsilon(0.0).with_max_iterations(6).minimize(
        x, objectives.Quadratic(), lambda p: rows.append([p.niter, p.fx, p.gnorm]) and False, ctx=ctx)
    out["rows"] = rows
except R.LbfgsError as e:
    out["error"] = str(e)
# every rank tells the others whether it has a context BEFORE anybody uses one: in a deployment that is the caller's job too
# (dist.sharded_context(kind="p2p") does it for P2P); the library's part is that ctx_create RETURNS on every rank
oks = [None] * world
dist.all_gather_object(oks, ctx is not None)
out["ranks_with_a_context"] = oks
if ctx is not None:
    ctx.close()
json.dump(out, open(os.path.join(os.environ["LBFGS_OUT"], f"rank{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


def _gated_world(tmp_path, world, n, extra_env):
    script = tmp_path / "gated_worker.py"
    script.write_text(GATED_WORKER)
    log = tmp_path / "protocol"
    env = dict(os.environ, LBFGS_ROOT=ROOT, LBFGS_OUT=str(tmp_path), OMP_NUM_THREADS="1", LBFGS_MOCK_RCCL="1", LBFGS_TEST_N=str(n),
               LBFGS_HIP_RCCL_RESIDENT="1", LBFGS_MOCK_FAKE_KERNEL_TIMES="1", LBFGS_MOCK_PROTOCOL_LOG=str(log))
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    outs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world) if (tmp_path / f"rank{k}.json").exists()]
    logs = [open(f"{log}.rank{k}").read().split() for k in range(world) if os.path.exists(f"{log}.rank{k}")]
    return r, outs, logs


@pytest.mark.parametrize("name,env,n,want", [
    ("all_fine", {}, 5000, dict(resident=True, errors=[])),
    # the trial comes back wrong on rank 1 only, AFTER the handshake passed everywhere: every rank must take the kernel-per-step form
    ("trial_fails_on_one_rank", {"LBFGS_HIP_RESIDENT_FAULT": "-1", "LBFGS_MOCK_FAULT_RANK": "1"}, 5000, dict(resident=False, errors=[])),
    ("streams_share_a_queue_on_one_rank", {"LBFGS_MOCK_HANDSHAKE_FAIL_RANK": "2"}, 5000, dict(resident=False, errors=[])),
    # shard_range(600, r, 3) = [0,256) [256,512) [512,600): fine; n = 500 leaves rank 2 EMPTY -- it cannot take the persistent kernel
    ("empty_shard_on_the_last_rank", {}, 500, dict(resident=False, errors=[])),
], ids=lambda v: v if isinstance(v, str) else None)
def test_gated_exchange_preparation_agrees_across_ranks(name, env, n, want, tmp_path):
    """The product's collective skeleton (ext_protocol.h) on three rank processes of the test double: whatever ONE rank finds,
    EVERY rank ends with the same launch form, and every rank made the same sequence of collectives (the double tags each
    all-reduce with what its caller thinks it is; a mismatch is an error)."""
    r, outs, logs = _gated_world(tmp_path, 3, n, env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(outs) == 3 and all("error" not in o for o in outs), outs
    assert [o["resident_form"] for o in outs] == [want["resident"]] * 3
    assert all(o["rows"] == outs[0]["rows"] and len(o["rows"]) == 6 for o in outs)
    assert len(logs) == 3 and logs[0] == logs[1] == logs[2] and len(logs[0]) > 0
    if not want["resident"]:
        assert "gated RCCL exchange is not used" in r.stderr


@pytest.mark.parametrize("fault,rank,says", [("-2", 0, "second stream"), ("-3", 2, "before the gated exchange's trial")])
def test_a_rank_local_failure_in_the_preparation_leaves_no_peer_waiting(fault, rank, says, tmp_path):
    """Round-5 advice: a rank that fails LOCALLY between the collectives of lbfgs_hip_ctx_create (stream creation, allocation, the
    trial's launch) used to return at once and leave its peers inside ncclAllReduce for ever.  Now it votes "bad", takes part in
    every collective that follows -- the trial's all-reduce included -- and only then returns its error: the other ranks get
    their contexts (kernel per step) and RETURN.  Here: world 3, the failure injected on one rank."""
    script = tmp_path / "gated_worker.py"
    script.write_text(GATED_WORKER.replace(
        "    x = np.zeros(ctx.n_local)",
        "    oks0 = [None] * world\n    dist.all_gather_object(oks0, True)\n    if not all(oks0):\n        raise R.LbfgsError(-1, 'a peer has no context: not optimising')\n    x = np.zeros(ctx.n_local)", 1).replace(
        "except R.LbfgsError as e:\n    out[\"error\"] = str(e)",
        "except R.LbfgsError as e:\n    out[\"error\"] = str(e)\n    if ctx is None:\n        oks0 = [None] * world\n        dist.all_gather_object(oks0, False)", 1))
    log = tmp_path / "protocol"
    env = dict(os.environ, LBFGS_ROOT=ROOT, LBFGS_OUT=str(tmp_path), OMP_NUM_THREADS="1", LBFGS_MOCK_RCCL="1", LBFGS_TEST_N="5000",
               LBFGS_HIP_RCCL_RESIDENT="1", LBFGS_MOCK_FAKE_KERNEL_TIMES="1", LBFGS_MOCK_PROTOCOL_LOG=str(log),
               LBFGS_HIP_RESIDENT_FAULT=fault, LBFGS_MOCK_FAULT_RANK=str(rank))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    outs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(3)]
    for o in outs:
        if o["rank"] == rank:
            assert "injected" in o["error"] and says in o["error"], o
        else:  # a context, of the kernel-per-step form -- which it does not use, because a peer has none
            assert o.get("resident_form") is False and "a peer has no context" in o["error"], o
        assert o["ranks_with_a_context"] == [k != rank for k in range(3)]
    logs = [open(f"{log}.rank{k}").read().split() for k in range(3)]
    assert logs[0] == logs[1] == logs[2] and len(logs[0]) >= 2      # the same collectives on every rank, the failing one included
