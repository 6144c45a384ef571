"""N > 1 on the real HIP library with ONE GPU: two processes share GPU 0, each holds a contiguous
shard, scalars are closed through the callback communicator (gloo).  Plus the RCCL code path with a
1-rank communicator (dlopen, ncclCommInitRank, in-place ncclAllReduce on the compute stream)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import rust_lbfgs_amd as R
from rust_lbfgs_amd import _ffi
from rust_lbfgs_amd.math import DeviceVec
from tests.test_distributed_cpu import compare_sharded_fuzz, oracle_rows, run_world
from tests.test_gpu_parity import product_library  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [
    dict(name="quadratic", n=200_003, m=7, iters=25, objective="quadratic"),
    dict(name="owlqn_straddle", n=30_001, m=6, iters=20, objective="logistic", owl=[0.5, 7000, 29000]),
    dict(name="host_closure", n=1000, m=4, iters=12, objective="closure"),
], ids=lambda c: c["name"])
@pytest.mark.parametrize("kind", ["callback", "p2p", "p2p-host"])
def test_two_processes_one_gpu(case, kind, tmp_path, monkeypatch):
    """kind = "p2p": the direct-exchange communicator (IPC mailboxes, tagged granules) between two
    processes whose "peer" is the same GPU -- the protocol is identical to the 8-GPU xGMI case.
    kind = "p2p-host": the same exchange with the mailboxes in host-coherent shared memory (the second placement:
    what a machine falls back to when device memory cannot be mapped between its GPUs)."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("covered by tests/test_distributed_cpu.py")
    monkeypatch.setenv("LBFGS_WORKER_PRODUCT", "1")
    monkeypatch.setenv("LBFGS_COMM_KIND", kind)
    outs = run_world(case, 2, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    if kind != "callback":
        assert [o["placement"] for o in outs] == ["host" if kind == "p2p-host" else "device"] * 2
    assert outs[0]["rows"] == outs[1]["rows"]
    assert len(outs[0]["rows"]) == len(ref_rows)
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6), (got, ref)
    x = np.concatenate([np.array(o["x"]) for o in outs])
    assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)


def test_p2p_falls_back_to_host_mailboxes_when_ipc_mapping_fails(tmp_path, monkeypatch):
    """hipIpcOpenMemHandle of a peer's device memory refused (simulated: LBFGS_HIP_TEST_FAIL_IPC_OPEN) on every rank: the
    ranks decide TOGETHER to drop the device-placed mailboxes and come up with host-placed ones -- same trajectory, same
    bits as the device placement gives (the exchange adds the ranks' totals in rank order either way)."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    monkeypatch.setenv("LBFGS_WORKER_PRODUCT", "1")
    monkeypatch.setenv("LBFGS_COMM_KIND", "p2p")
    case = dict(name="quadratic", n=200_003, m=7, iters=25, objective="quadratic")
    dev = run_world(case, 2, tmp_path)
    assert [o["placement"] for o in dev] == ["device", "device"]
    monkeypatch.setenv("LBFGS_HIP_TEST_FAIL_IPC_OPEN", "1")
    host = run_world(case, 2, tmp_path)
    assert [o["placement"] for o in host] == ["host", "host"]
    assert host[0]["rows"] == host[1]["rows"] == dev[0]["rows"]          # bitwise
    assert host[0]["x"] == dev[0]["x"] and host[1]["x"] == dev[1]["x"]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("lbfgs_hip_mbox_")]   # the segments' names are gone


def test_rccl_single_rank_communicator():
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs RCCL")
    from rust_lbfgs_amd.dist import CommSpec

    L = _ffi.load()
    _ffi.torch_before_rccl()  # as dist.rccl_comm() does: torch initialises before librccl is opened
    buf = (C.c_char * 128)()
    assert L.lbfgs_hip_rccl_unique_id(buf) == 0, L.lbfgs_hip_last_error(None)
    n = 100_001
    shard = _ffi.Shard(0, 1, n, 0, n)
    with R.Context(n, shard=shard, comm=CommSpec(_ffi.COMM_RCCL, unique_id=buf)) as ctx:
        x, y = DeviceVec(ctx), DeviceVec(ctx)
        x.fill(2.0); y.fill(0.5)
        assert x.vecdot(y) == float(n)            # 1-rank all-reduce is the identity
        assert x.vec2norm() == np.sqrt(4.0 * n)
        nred, ms = ctx.prof_read(_ffi.K_COMM)
        x.free(); y.free()


@pytest.mark.parametrize("kind", ["p2p", "p2p-host"])
@pytest.mark.parametrize("world,n,grid", [(2, 1_300_003, 80), (3, 1_500_001, 48), (2, 3_400_001, 32)])
def test_resident_two_loop_with_the_p2p_exchange(world, n, grid, kind, tmp_path, monkeypatch):
    """The on-chip-resident two-loop kernel (resident.h) under the P2P communicator: workgroup 0 of every rank closes
    each of the kernel's 2m hand-offs across the ranks through the mailboxes and broadcasts the global total to its
    other workgroups.  That path is meant for ranks that own their GPU (`exclusive_device`); here two / three ranks
    share ONE, so each is given a third / a fifth of the CUs (LBFGS_HIP_RESIDENT_GRID) so that all of them are resident
    together.  Checked against the single-rank oracle, and every rank must really have run the resident kernel.  The
    third case is HYBRID: each rank's shard exceeds what its 32 workgroups hold, so part of q stays in
    HBM (the form that shards of more than 1.25e7 elements take on a whole GPU)."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    monkeypatch.setenv("LBFGS_WORKER_PRODUCT", "1")
    monkeypatch.setenv("LBFGS_COMM_KIND", kind)
    monkeypatch.setenv("LBFGS_TEST_EXCLUSIVE_DEVICE", "1")
    monkeypatch.setenv("LBFGS_HIP_RESIDENT_GRID", str(grid))
    case = dict(name="quadratic_resident", n=n, m=6, iters=14, objective="quadratic")
    outs = run_world(case, world, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    for o in outs:
        assert o["placement"] == ("host" if kind == "p2p-host" else "device")
        assert o["rows"] == outs[0]["rows"]
        assert o["resident"] >= 10, o["resident"]
        if n > 3_000_000:
            assert 0 < o["resident_elements"] < o["hi"] - o["lo"], o["resident_elements"]   # hybrid
        else:
            assert o["resident_elements"] == o["hi"] - o["lo"]
    assert len(outs[0]["rows"]) == len(ref_rows)
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6), (got, ref)
    x = np.concatenate([np.array(o["x"]) for o in outs])
    assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)


@pytest.mark.parametrize("kind", ["p2p", "p2p-host"])
def test_resident_p2p_on_a_near_production_grid(kind, tmp_path, monkeypatch):
    """The persistent kernel with the exchange inside its hand-offs on (almost) the grid of the 8-GPU run: rank 0 runs it
    on 224 of this GPU's 256 CUs -- 224 workgroups polling one another, workgroup 0 exchanging with the peer -- over a
    shard of 3e6 elements, while a second rank with a sliver of the vector takes 8 of the CUs next to it.
    (Why not 240 + 8: measured with tools/near_grid_probe.py, two PROCESSES' kernels that each want a whole CU per workgroup
    are co-resident on one MI355X up to 232 + 8 and 247 + 1 workgroups, but not at 240 + 8 or 240 + 1 -- the dispatcher does
    not hand the last CUs of an XCD to a second process's kernel; profiles/r03_near_grid_probe.log.  One rank per GPU, the
    deployment this path is for, never shares.)"""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    monkeypatch.setenv("LBFGS_WORKER_PRODUCT", "1")
    monkeypatch.setenv("LBFGS_COMM_KIND", kind)
    monkeypatch.setenv("LBFGS_TEST_EXCLUSIVE_DEVICE", "1")
    monkeypatch.setenv("LBFGS_TEST_RESIDENT_GRIDS", "224,8")
    n, cut = 3_000_000 + 40_960 + 5, 3_000_000
    monkeypatch.setenv("LBFGS_TEST_BOUNDS", json.dumps([0, cut, n]))
    case = dict(name="quadratic_resident_224", n=n, m=6, iters=12, objective="quadratic")
    outs = run_world(case, 2, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    assert [o["hi"] - o["lo"] for o in outs] == [cut, n - cut]
    for o in outs:
        assert o["err"] == 0, (o["rank"], o["errmsg"])
        assert o["rows"] == outs[0]["rows"]
        assert o["resident"] >= 9 and o["resident_elements"] == o["hi"] - o["lo"], (o["resident"], o["resident_elements"])
    assert len(outs[0]["rows"]) == len(ref_rows)
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6), (got, ref)
    x = np.concatenate([np.array(o["x"]) for o in outs])
    assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)


@pytest.mark.parametrize("kind", ["p2p", "p2p-host"])
@pytest.mark.parametrize("case", [
    dict(name="quadratic3_mixed", n=1000, m=4, iters=15, objective="quadratic"),
    dict(name="owlqn3_mixed", n=1000, m=5, iters=15, objective="logistic", owl=[0.5, 100, 900]),
], ids=lambda c: c["name"])
def test_ranks_may_take_different_two_loop_paths(case, kind, tmp_path, monkeypatch):
    """Whether a rank runs the two-loop as the persistent resident kernel or as a kernel per step is decided per rank
    (an EMPTY shard, for one, has nothing to keep on the chip).  Both forms close the same sequence of reductions with
    the same number of values each -- under OWL-QN too, where the last four sums travel together in either form -- so
    ranks that decide differently still meet in every P2P exchange.  world = 3 on one GPU, shards of 512, 488 and 0
    elements: ranks 0 and 1 run the resident kernel (a few workgroups each), rank 2 the kernel-per-step path."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    monkeypatch.setenv("LBFGS_WORKER_PRODUCT", "1")
    monkeypatch.setenv("LBFGS_COMM_KIND", kind)
    monkeypatch.setenv("LBFGS_TEST_EXCLUSIVE_DEVICE", "1")
    monkeypatch.setenv("LBFGS_HIP_RESIDENT_GRID", "48")
    outs = run_world(case, 3, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    assert [o["hi"] - o["lo"] for o in outs] == [512, 488, 0]
    assert outs[0]["resident"] >= 10 and outs[1]["resident"] >= 10 and outs[2]["resident"] == 0
    assert outs[0]["rows"] == outs[1]["rows"] == outs[2]["rows"]
    assert len(outs[0]["rows"]) == len(ref_rows)
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6), (got, ref)


@pytest.mark.parametrize("seed,vector_free", [(3, False), (12, True), (33, False), (41, True)])
def test_random_configurations_two_processes_p2p(seed, vector_free, tmp_path, monkeypatch):
    """Random configurations sharded over two processes on one GPU with the in-kernel P2P exchange."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("covered by tests/test_distributed_cpu.py")
    monkeypatch.setenv("LBFGS_WORKER_PRODUCT", "1")
    monkeypatch.setenv("LBFGS_COMM_KIND", "p2p")
    compare_sharded_fuzz(seed, tmp_path, vector_free)


@pytest.mark.parametrize("kind", ["p2p", "p2p-host", "callback"])
def test_three_processes_one_gpu_with_an_empty_shard(kind, tmp_path, monkeypatch):
    """world = 3 on one GPU: shards of 512, 488 and 0 elements (an EMPTY shard still takes part in every
    reduction and in the P2P exchange)."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("covered by tests/test_distributed_cpu.py")
    monkeypatch.setenv("LBFGS_WORKER_PRODUCT", "1")
    monkeypatch.setenv("LBFGS_COMM_KIND", kind)
    case = dict(name="quadratic3", n=1000, m=4, iters=15, objective="quadratic")
    outs = run_world(case, 3, tmp_path)
    ref_rows, ref_x = oracle_rows(case)
    assert [o["hi"] - o["lo"] for o in outs] == [512, 488, 0]
    assert outs[0]["rows"] == outs[1]["rows"] == outs[2]["rows"]
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6)


@pytest.mark.parametrize("world,launch,kind", [(1, "plain", "auto"), (2, "plain", "auto"), (3, "torchrun", "callback"),
                                               (2, "torchrun", "p2p")])
def test_bench_contract_line(world, launch, kind, tmp_path):
    """bench.py end to end at a tiny size: exactly ONE JSON line on stdout with the contract's keys, also when a
    rank holds an EMPTY shard (world 3, n = 1000) -- every rank must take the same collective decisions.  N > 1 is
    started BOTH ways: plain `python bench.py --gpus 2` (the supervisor launches its own ranks; on this one-GPU box
    the p2p leg works and the rccl leg fails -- two ranks on one device -- and the line still comes) and through
    torch.distributed.run (every rank a supervisor)."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--gpus", str(world), "--steps", "6", "--warmup", "12", "--dim", "1000" if world > 1 else "200000",
            "--cpu-n", "20000", "--device", "0", "--repeats", "3", "--leg-timeout", "150"]
    if launch == "plain":
        cmd = [sys.executable, "bench.py"] + args + ["--comm", kind]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(29600 + world), "bench.py"] + args + ["--comm", kind, "--pg-backend", "gloo"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == world and j["steps"] == 6 and j["warmup"] == 12 and j["dtype"] == "f64"
    assert j["vs_baseline"] is None and j["value"] > 0 and "workload" in j["config"]
    assert len(j["config"]["repeats_iters_per_sec"]) == 3
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0
    if world == 1:
        c = j["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
        assert c["host_cores_total"] >= 1 and c["host_cpu"]
    else:
        legs = j["config"]["legs"]
        assert any(v["status"] == "ok" for v in legs.values()), legs
        if launch == "plain":
            # one GPU: both mailbox placements work between the two processes; RCCL cannot put two ranks on one device, so its
            # PROBE fails and it is never measured
            probes = j["config"]["probes"]
            assert legs["p2p"]["status"] == "ok" and legs["p2p-host"]["status"] == "ok", legs
            assert probes["rccl"]["status"] != "ok" and "rccl" not in legs, (probes, legs)
            assert probes["p2p"]["mailboxes"] == "device" and probes["p2p-host"]["mailboxes"] == "host", probes
            # (round 4) every leg says what it spanned and what an exchange cost -- here with a kernel per two-loop step: the
            # reducing kernel's last workgroup closes each reduction across the two ranks and times it
            for name, placement in (("p2p", "device"), ("p2p-host", "host")):
                leg = legs[name]
                assert leg["ranks_seen"] == 2 and leg["mailbox_placement"] == placement, leg
                assert leg["exchanges_per_two_loop"] >= 10 and leg["exchange_us_mean"] > 0.0, leg  # (2 * bound, bound <= m = 10)
                # (round 6, ABI 5) ... and how the exchanges are DISTRIBUTED: quantiles from the device's histogram, the worst one
                assert 0.0 < leg["exchange_us_p50"] <= leg["exchange_us_p99"] <= leg["exchange_us_max"] + 0.25, leg
                assert leg["local_wait_us_max"] >= 0.0
        # ... and an N > 1 line carries the CPU baseline (timed by the supervisor, which touches no GPU)
        assert j["cpu_baseline"]["value"] > 0 and "supervisor" in j["cpu_baseline"]["where"]


def test_bench_survives_a_hung_leg(tmp_path):
    """`python bench.py --gpus 2` with a first leg that never returns: killed at --leg-timeout, the p2p leg still
    produces the line, exit code 0."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "12", "--dim", "1000", "--device", "0",
           "--repeats", "2", "--leg-timeout", "25", "--no-vector-free"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LBFGS_BENCH_LEGS="hang,p2p")
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    legs = json.loads(lines[0])["config"]["legs"]
    assert "timed out" in legs["hang"]["status"] and legs["p2p"]["status"] == "ok"


TIMEOUT_WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.environ["LBFGS_ROOT"])
import torch.distributed as dist
import rust_lbfgs_amd as R
from rust_lbfgs_amd import _ffi
from rust_lbfgs_amd import dist as D
from rust_lbfgs_amd.math import DeviceVec
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 100_000
lo, hi = D.shard_range(n, rank, world)
ctx = R.Context(n, device=0, shard=_ffi.Shard(rank, world, n, lo, hi - lo), comm=D.p2p_comm(0, timeout_s=0.3))
u = DeviceVec(ctx)
u.fill(1.0)
out = dict(rank=rank)
out["together"] = u.vecdot(u)                      # both ranks take part: n
dist.barrier()
if rank == 0:                                      # rank 1 stays away from the next two reductions
    t0 = time.time()
    for name, fn in (("in_kernel", lambda: u.vecdot(u)),
                     ("standalone", lambda: (ctx.set_scalars(200, [1.0]), ctx.check(ctx._L.lbfgs_hip_scalars_allreduce(ctx._h, 200, 1)), ctx.scalars(200))[-1])):
        try:
            fn()
            out[name] = "no error"
        except R.LbfgsError as e:
            out[name] = [e.code, str(e)]
    out["seconds"] = time.time() - t0
dist.barrier()
u.free()
ctx.close()
json.dump(out, open(os.path.join(os.environ["LBFGS_OUT"], f"rank{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("placement", ["device", "host"])
def test_p2p_exchange_times_out_instead_of_hanging(placement, tmp_path, monkeypatch):
    """A peer that never arrives: the bounded spin gives up after p2p_timeout_s and the NEXT scalar read fails with
    LBFGS_HIP_ERR_COMM -- through the host mirror (in-kernel exchange) and through the copy path alike, with the mailboxes
    in device memory and in host shared memory."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    monkeypatch.setenv("LBFGS_HIP_P2P_MAILBOX", placement)
    import subprocess
    import sys

    from tests.test_distributed_cpu import ROOT, _free_port

    script = tmp_path / "worker.py"
    script.write_text(TIMEOUT_WORKER)
    env = dict(os.environ, LBFGS_ROOT=ROOT, LBFGS_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.load(open(tmp_path / "rank0.json"))
    assert out["together"] == 100_000.0
    assert out["in_kernel"][0] == _ffi.HIP_ERR_COMM and "timed out" in out["in_kernel"][1], out
    assert out["standalone"][0] == _ffi.HIP_ERR_COMM, out
    assert out["seconds"] < 30.0


def test_bench_p2p_leg_with_the_persistent_kernel(tmp_path):
    """The shape of the driver's multi-GPU run as far as ONE GPU can rehearse it: `python bench.py --gpus 2 --comm p2p` with
    `--exclusive-device 1` (ranks own "their" GPU: here each gets 32 workgroups of it), so the leg's start-up self-test
    and the timed iterations run the persistent two-loop kernel with the exchange inside its hand-offs -- in its HYBRID
    form, since 5e6 elements per rank exceed what 32 workgroups hold."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--device", "0", "--comm", "p2p", "--exclusive-device", "1", "--dim",
           "10000000", "--steps", "5", "--warmup", "12", "--repeats", "1", "--no-vector-free", "--leg-timeout", "200"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LBFGS_HIP_RESIDENT_GRID="32")
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["config"]["legs"]["p2p"]["status"] == "ok", j["config"]["legs"]
    r = j["roofline"]
    assert r["kernel"].startswith("two_loop_resident_kernel") and 0 < r["resident_elements"] < j["config"]["n_local_rank0"]
    assert "self-test" not in p.stderr
    # round 4: the line is attributable -- what the communicator spanned, what an exchange cost (measured on the device by
    # workgroup 0 of the persistent kernel), how many exchanges a two-loop made, and the CPU baseline on an N > 1 line
    leg, ci = j["config"]["legs"]["p2p"], j["config"]["comm_info"]
    assert leg["ranks_seen"] == 2 and ci["kind"] == "p2p" and ci["peers_device"] == 1 and ci["mailbox_placement"] == "device"
    assert leg["exchanges_per_two_loop"] == pytest.approx(20.0)       # m = 10, history full: one exchange per hand-off
    assert 0.0 < leg["exchange_us_mean"] < 1000.0 and r["exchange_us_mean"] == leg["exchange_us_mean"]
    assert leg["local_wait_us_mean"] is not None and r["exchanges_per_two_loop"] == pytest.approx(20.0)
    # (round 6, ABI 5) the distribution behind the mean, from workgroup 0's fire-and-forget histogram updates inside the persistent kernel
    assert 0.0 < leg["exchange_us_p50"] <= leg["exchange_us_p99"] <= leg["exchange_us_max"] + 0.25 and leg["exchange_us_p50"] < 1000.0, leg
    assert ci["exchange_hist_counted"] > 0 and leg["local_wait_us_max"] >= leg["local_wait_us_mean"]
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] == 1


def test_world_8_on_one_gpu_as_processes_times_threads(tmp_path):
    """The metric's own world (8) on the one GPU of the test box: the pool admits six GPU processes per card, so the eight
    ranks are 4 processes x 2 host threads (tools/eight_ranks_one_gpu.py) -- each with its own context, stream, shard and
    mailbox; same-process peers reach each other's device-placed mailbox through the pointer, the others through HIP IPC.
    Whole optimisations against the single-rank ORACLE (on-chip and hybrid shard sizes, quadratic and OWL-QN: 24 workgroups
    of the persistent kernel per rank, the exchange inside its hand-offs), then bench.py's own make_context + measure at a
    reduced size: eight mailboxes seen by every rank, 2m exchanges per two-loop, rank-ordered sums of eight, the seven
    full shards and the short last one."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU (the CPU suite runs world 8 through bench.py's supervisor on the test double)")
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "bench_eight_ranks_sharing_one_gpu.json")
    keep = out + ".full_size"
    if os.path.exists(out):  # (a full-size record of the same tool, made by hand: keep it)
        os.replace(out, keep)
    try:
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "eight_ranks_one_gpu.py"), "--dim", "20000003", "--hist", "6",
                            "--steps", "20", "--repeats", "2", "--legs", "p2p,p2p-host"], cwd=root, capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        rec = json.load(open(out))
    finally:
        if os.path.exists(keep):
            os.replace(keep, out)
    assert rec["ok"] and rec["exit_codes"] == [0, 0, 0, 0]
    res = rec["result"]
    assert len(res["trajectories"]) == 6 and all(t["ok"] and t["ranks_seen"] == [8] * 8 for t in res["trajectories"])
    assert any(t["on_chip_elements"][0] < t["shard_elements"][0] for t in res["trajectories"])       # a hybrid case ...
    assert any(t["on_chip_elements"][0] == t["shard_elements"][0] for t in res["trajectories"])      # ... and an on-chip one
    for leg, placement in (("p2p", "device"), ("p2p-host", "host")):
        line = res["legs"][leg]["line"]
        ci = line["config"]["comm_info"]
        assert line["n_gpus"] == 8 and line["value"] > 0 and ci["ranks_seen"] == 8 and ci["mailbox_placement"] == placement
        assert ci["peers_device"] + ci["peers_host"] == 7 and ci["exchanges_per_two_loop"] == pytest.approx(12.0)
        assert ci["exchange_us_mean"] and ci["resident_fallbacks"] == 0
        assert line["roofline"]["kernel"].startswith("two_loop_resident_kernel")
        assert "NOT RCCL" in line["metric"]


@pytest.mark.parametrize("n,m,owl", [(1_300_003, 6, None), (20_000_003, 5, None), (400_001, 6, (0.5, 30_000, 390_000))],
                         ids=["on_chip", "hybrid", "owlqn"])
def test_rccl_gated_exchange_under_the_persistent_kernel(n, m, owl, monkeypatch):
    """RCCL under the persistent two-loop kernel (round 5): ncclAllReduce only exists as a host-enqueued kernel, so the host
    enqueues one per hand-off on a SECOND stream, each behind a gate kernel that waits for the persistent kernel's flag; the
    kernel stores its sums into an uncached slot, raises flag A, and reads the reduced sums back once flag B carries the same
    epoch -- q never leaves the chip (csrc/stream.h ext_exchange, lbfgs_hip.hip enqueue_gated_chain).  One GPU can only host
    a 1-rank communicator, whose all-reduce is the identity: what is checked here is the whole mechanism -- gates, epochs,
    the ring of slots, both streams, the mirror, the counters -- on whole optimisations against the ORACLE and against the same
    communicator with a kernel per step (LBFGS_HIP_RCCL_RESIDENT=0), and that the kernel really ran resident."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs RCCL and the GPU")
    from oracle import oracle as O
    from rust_lbfgs_amd import objectives
    from rust_lbfgs_amd.dist import CommSpec

    L = _ffi.load()
    _ffi.torch_before_rccl()
    iters = 14

    def run(resident):
        monkeypatch.setenv("LBFGS_HIP_RCCL_RESIDENT", "1" if resident else "0")
        buf = (C.c_char * 128)()
        assert L.lbfgs_hip_rccl_unique_id(buf) == 0, L.lbfgs_hip_last_error(None)
        spec = CommSpec(_ffi.COMM_RCCL, unique_id=buf)
        spec.c.exclusive_device = 1
        rows = []
        with R.Context(n, shard=_ffi.Shard(0, 1, n, 0, n), comm=spec) as ctx:
            b = R.lbfgs().with_m(m).with_epsilon(0.0).with_max_iterations(iters)
            if owl:
                b = b.with_orthantwise(*owl)
            x = np.zeros(n)
            b.minimize(x, objectives.Logistic() if owl else objectives.Quadratic(),
                       lambda p: rows.append((p.niter, p.neval, p.ncall, p.fx, p.xnorm, p.gnorm, p.step)) and False, ctx=ctx)
            ci = ctx.comm_info()
            assert sum(ci["exchange_hist"]["two_loop"]) == ci["timed_exchanges"]["two_loop"]   # (ABI 5: every timed exchange is in a bin)
            assert ci["exchange_us_max"]["two_loop"] * ci["timed_exchanges"]["two_loop"] >= ci["exchange_us"]["two_loop"] * 0.999
            stats = dict(resident=ctx.resident_two_loops(), on_chip=ctx.resident_elements(), two_loops=ci["two_loops"],
                         exchanges=ci["two_loop_exchanges"], timed=ci["timed_exchanges"]["two_loop"],
                         us=ci["exchange_us"]["two_loop"], ranks_seen=ci["ranks_seen"], fallbacks=ci["resident_fallbacks"])
        return x, rows, stats

    xg, rg, sg = run(True)
    xs, rs, ss = run(False)
    # the gated form really ran: every two-loop with history resident, 2*bound (+1 under OWL-QN) exchanges each, timed on the device
    assert sg["resident"] == iters - 1 and ss["resident"] == 0 and sg["fallbacks"] == 0 and sg["ranks_seen"] == 1
    assert (sg["on_chip"] < n) == (n > 12_582_912)
    assert sg["timed"] == sg["exchanges"] > 2 * (iters - 1 - m) * m and sg["us"] > 0.0
    # a 1-rank all-reduce is the identity: the gated run is the per-step run up to the two launch forms' summation orders
    # (the persistent kernel's workgroups own other elements than the streaming kernels' do) ...
    assert len(rg) == len(rs)
    for a, b in zip(rg, rs):
        assert a[:3] == b[:3]
        for u, v in zip(a[3:], b[3:]):
            assert abs(u - v) <= 1e-11 * max(abs(u), 1e-6), (a, b)
    # ... and both are the oracle's run
    ro, xo = [], np.zeros(n)
    bo = O.lbfgs().with_m(m).with_epsilon(0.0).with_max_iterations(iters)
    if owl:
        bo = bo.with_orthantwise(*owl)
    bo.minimize(xo, O.logistic() if owl else O.quadratic(),
                lambda p: ro.append((p["niter"], p["neval"], p["ncall"], p["fx"], p["xnorm"], p["gnorm"], p["step"])) and False)
    assert len(ro) == len(rg)
    for a, b in zip(ro, rg):
        assert a[:3] == b[:3]
        for u, v in zip(a[3:], b[3:]):
            assert abs(u - v) <= 1e-9 * max(abs(u), 1e-6), (a, b)
    assert np.max(np.abs(xg - xo)) <= 1e-9 * max(np.max(np.abs(xo)), 1e-12)
    print(f"gated RCCL exchange, n={n}: {sg['exchanges']} exchanges in {sg['two_loops']} two-loops, {sg['us'] / sg['timed']:.2f} us each "
          f"(1-rank communicator: gate + post, no all-reduce kernel), {sg['on_chip']} of {n} elements on the chip")


def test_rccl_gated_exchange_is_proven_at_context_creation_or_not_used(monkeypatch, capfd):
    """lbfgs_hip_ctx_create tries the gated exchange once on the context's own communicator -- an all-reduce enqueued on the second
    stream, awaited by a kernel of one workgroup per CU on all but eight CUs -- and the ranks agree on the outcome (a sum over
    the communicator): if ANY rank reports a failure, every rank takes the kernel-per-step form and says so.  The failure is
    injected here (LBFGS_HIP_RESIDENT_FAULT=-1: this rank reports its self-test as failed after running it); the run that follows
    must be the per-step run of the same communicator, and the context without the fault must run gated."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs RCCL and the GPU")
    from rust_lbfgs_amd import objectives
    from rust_lbfgs_amd.dist import CommSpec

    L = _ffi.load()
    _ffi.torch_before_rccl()
    n, m, iters = 1_300_003, 6, 10

    def run(fault):
        if fault:
            monkeypatch.setenv("LBFGS_HIP_RESIDENT_FAULT", "-1")
        else:
            monkeypatch.delenv("LBFGS_HIP_RESIDENT_FAULT", raising=False)
        monkeypatch.setenv("LBFGS_HIP_RCCL_RESIDENT", "1")
        buf = (C.c_char * 128)()
        assert L.lbfgs_hip_rccl_unique_id(buf) == 0, L.lbfgs_hip_last_error(None)
        spec = CommSpec(_ffi.COMM_RCCL, unique_id=buf)
        spec.c.exclusive_device = 1
        rows = []
        with R.Context(n, shard=_ffi.Shard(0, 1, n, 0, n), comm=spec) as ctx:
            x = np.zeros(n)
            R.lbfgs().with_m(m).with_epsilon(0.0).with_max_iterations(iters).minimize(
                x, objectives.Quadratic(), lambda p: rows.append((p.niter, p.neval, p.fx, p.gnorm)) and False, ctx=ctx)
            return x, rows, ctx.resident_two_loops(), ctx.comm_info()

    capfd.readouterr()
    xf, rf, resident_f, cif = run(True)
    err = capfd.readouterr().err
    assert "self-test" in err and "1 of 1 ranks" in err and "kernel per step" in err, err
    assert resident_f == 0 and cif["two_loops"] == iters - 1
    xg, rg, resident_g, cig = run(False)
    assert "self-test" not in capfd.readouterr().err
    assert resident_g == iters - 1 and cig["resident_fallbacks"] == 0
    assert len(rf) == len(rg) == iters
    for a, b in zip(rf, rg):
        assert a[:2] == b[:2]
        for u, v in zip(a[2:], b[2:]):
            assert abs(u - v) <= 1e-11 * max(abs(u), 1e-6), (a, b)
    assert np.max(np.abs(xf - xg)) <= 1e-11 * np.max(np.abs(xg))


def test_rccl_gated_exchange_is_opt_in_and_a_local_failure_is_returned_after_the_collectives(monkeypatch):
    """Round-5 advice.  (1) The gated exchange has never run with more than one rank, so the library's default under RCCL is
    the kernel-per-step form: LBFGS_HIP_RCCL_RESIDENT=1 opts in (bench.py's "rccl" leg does, inside a child job with a timeout).
    (2) lbfgs_hip_ctx_create's preparation of the gated exchange is COLLECTIVE -- agree, warm-up all-reduce, trial, agree -- and a
    rank that fails locally between those steps (stream creation, allocation, the trial's launch) must not return before the last
    of them: its peers would wait inside ncclAllReduce for ever.  Injected here on the only rank a single GPU can host: the
    context is refused with the local error, after every collective has been made (it does not hang, and the next context on
    the same device works)."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs RCCL and the GPU")
    from rust_lbfgs_amd import objectives
    from rust_lbfgs_amd.dist import CommSpec

    L = _ffi.load()
    _ffi.torch_before_rccl()
    n, m, iters = 1_300_003, 6, 8

    def make():
        buf = (C.c_char * 128)()
        assert L.lbfgs_hip_rccl_unique_id(buf) == 0, L.lbfgs_hip_last_error(None)
        spec = CommSpec(_ffi.COMM_RCCL, unique_id=buf)
        spec.c.exclusive_device = 1
        return R.Context(n, shard=_ffi.Shard(0, 1, n, 0, n), comm=spec)

    def run():
        with make() as ctx:
            x = np.zeros(n)
            R.lbfgs().with_m(m).with_epsilon(0.0).with_max_iterations(iters).minimize(x, objectives.Quadratic(), lambda p: False, ctx=ctx)
            return ctx.resident_two_loops()

    monkeypatch.delenv("LBFGS_HIP_RCCL_RESIDENT", raising=False)
    monkeypatch.delenv("LBFGS_HIP_RESIDENT_FAULT", raising=False)
    assert run() == 0                                   # default: a kernel per step
    monkeypatch.setenv("LBFGS_HIP_RCCL_RESIDENT", "1")
    assert run() == iters - 1                           # opted in: every two-loop with history ran as the persistent kernel
    for fault, says in (("-2", "second stream"), ("-3", "before the gated exchange's trial")):
        monkeypatch.setenv("LBFGS_HIP_RESIDENT_FAULT", fault)
        with pytest.raises(R.LbfgsError) as ei:
            make()
        assert "injected" in str(ei.value) and says in str(ei.value), str(ei.value)
    monkeypatch.delenv("LBFGS_HIP_RESIDENT_FAULT")
    assert run() == iters - 1
