"""The bench record's evidence fields (round 4): roofline.traffic is keyed to the REAL shard sizes and to the build that was
profiled; the N > 1 line says what its communicator spanned, what an exchange cost and carries the CPU baseline.
CPU only (the rank processes of the supervisor tests run on the test double of the C-ABI)."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import rust_lbfgs_amd  # noqa: E402,F401
from rust_lbfgs_amd import _build  # noqa: E402
from rust_lbfgs_amd.dist import shard_range  # noqa: E402

RESIDENT = "two_loop_resident_kernel<ER,NT>"


def _write(d, name, **kw):
    rec = dict(kernel="two_loop_resident_kernel", n_local=12500224, m=10, traffic_bytes_per_launch=4.1e9, _source="test",
               command="bench.py")
    rec.update(kw)
    with open(os.path.join(d, name), "w") as f:
        json.dump(rec, f)


def test_traffic_lookup_is_strict_about_size_m_and_kernel(tmp_path):
    d = str(tmp_path)
    _write(d, "pmc_traffic_a.json", n_local=12500000, build_id="aaaa")
    assert bench.traffic_lookup(12500224, 10, RESIDENT, build_id="aaaa", profiles_dir=d) == {}
    _write(d, "pmc_traffic_b.json", n_local=12500224, m=7, build_id="aaaa")
    assert bench.traffic_lookup(12500224, 10, RESIDENT, build_id="aaaa", profiles_dir=d) == {}
    _write(d, "pmc_traffic_c.json", kernel="stream_kernel", build_id="aaaa")
    assert bench.traffic_lookup(12500224, 10, RESIDENT, build_id="aaaa", profiles_dir=d) == {}
    _write(d, "pmc_traffic_d.json", build_id="aaaa")
    got = bench.traffic_lookup(12500224, 10, RESIDENT, build_id="aaaa", profiles_dir=d)
    assert got["traffic"] == pytest.approx(4.1) and got["traffic_is_current"] is True and got["traffic_build_id"] == "aaaa"
    assert got["traffic_file"].endswith("pmc_traffic_d.json")


def test_traffic_lookup_takes_only_passes_of_bench_py_itself(tmp_path):
    """tools/profile_configs.sh profiles tools/run_configs.py (OWL-QN logistic at n = 1e7, m = 6; damped Lennard-Jones at
    n = 3e6, m = 6) -- sizes `bench.py --dim ... --hist 6` can be asked to run with ANOTHER objective: those bytes are not this
    command's (round-4 advice)."""
    d = str(tmp_path)
    _write(d, "pmc_traffic_config5.json", n_local=3000000, m=6, build_id="aaaa", command="tools/run_configs.py --only config5",
           resident_elements=None)
    assert bench.traffic_lookup(3000000, 6, RESIDENT, build_id="aaaa", profiles_dir=d) == {}
    _write(d, "pmc_traffic_unstamped.json", n_local=3000000, m=6, build_id="aaaa", command=None)
    assert bench.traffic_lookup(3000000, 6, RESIDENT, build_id="aaaa", profiles_dir=d) == {}
    for path in __import__("glob").glob(os.path.join(ROOT, "profiles", "pmc_traffic_config*.json")):
        assert json.load(open(path)).get("command", "").startswith("tools/run_configs.py"), path


def test_traffic_lookup_names_a_stale_build_and_prefers_the_current_one(tmp_path):
    d = str(tmp_path)
    _write(d, "pmc_traffic_old.json", build_id="0ld0", traffic_bytes_per_launch=9e9)
    got = bench.traffic_lookup(12500224, 10, RESIDENT, build_id="new1", profiles_dir=d)
    assert got["traffic_is_current"] is False and got["traffic_build_id"] == "0ld0" and got["loaded_build_id"] == "new1"
    _write(d, "pmc_traffic_zz_unstamped.json")  # a file from before the ids existed: never "current"
    _write(d, "pmc_traffic_new.json", build_id="new1", traffic_bytes_per_launch=4e9)
    got = bench.traffic_lookup(12500224, 10, RESIDENT, build_id="new1", profiles_dir=d)
    assert got["traffic_is_current"] is True and got["traffic"] == pytest.approx(4.0)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_every_rank0_shard_of_the_metric_has_a_current_traffic_file(world):
    """BASELINE.json's metric at 1 / 2 / 4 / 8 GPUs: the rank-0 shard sizes are 1e8, 50 000 128, 25 000 192 and 12 500 224
    elements (dist.shard_range rounds to 256), and each must find committed counter passes taken at exactly that size with
    the build of this checkout -- otherwise a SCALE line would carry `traffic: null` (or a figure of another build)."""
    lo, hi = shard_range(100_000_000, 0, world)
    assert hi - lo == {1: 100_000_000, 2: 50_000_128, 4: 25_000_192, 8: 12_500_224}[world]
    got = bench.traffic_lookup(hi - lo, 10, RESIDENT, build_id=_build.hip_build_id())
    assert got.get("traffic"), f"no profiles/pmc_traffic*.json for n_local={hi - lo}, m=10 (tools/profile_round.sh)"
    assert got["traffic_is_current"], (f"{got['traffic_file']} was taken with build {got['traffic_build_id']}, the checked-out "
                                       f"sources hash to {_build.hip_build_id()}: re-take it (tools/profile_round.sh)")
    # the traffic is within 3 % of the kernel's byte model (4m+1 passes over the on-chip elements, 8m-1 over the rest): a
    # little above where the waiting workgroups' touches re-read lines (P = 8: 1.005), a little below where the hybrid form's
    # alternating sweep finds what it has just written still in the L2s (P = 4: 0.980)
    pm = json.load(open(os.path.join(ROOT, got["traffic_file"])))
    if pm.get("algorithmic_bytes_per_launch"):
        assert abs(pm["traffic_bytes_per_launch"] / pm["algorithmic_bytes_per_launch"] - 1.0) < 0.03


def test_every_traffic_file_names_its_build():
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_traffic*.json")))
    assert files
    for path in files:
        pm = json.load(open(path))
        assert pm.get("build_id"), f"{path} does not say which build it was taken with"


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("launcher", [False, True], ids=["plain_python", "torch_distributed_run"])
def test_the_n_gt_1_line_is_attributable(launcher):
    """The supervisor's line carries the CPU baseline (timed beside the legs), and per leg what the communicator spanned
    and how many exchanges a two-loop made (on the test double: the callback communicator, 2*bound + 2 all-reduces)."""
    env = dict(os.environ, LBFGS_BENCH_WORKER=os.path.join(ROOT, "tests", "support", "bench_on_mock.py"), OMP_NUM_THREADS="1")
    args = ["--gpus", "2", "--steps", "4", "--warmup", "12", "--dim", "3000", "--hist", "5", "--repeats", "2", "--no-vector-free",
            "--comm", "callback"]
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port" and "supervisor" in cb["where"]
    ci = j["config"]["comm_info"]
    assert ci["kind"] == "callback" and ci["world"] == 2 and ci["rank"] == 0
    leg = j["config"]["legs"]["callback"]
    # m = 5, history full: the test double's unfused recursion closes 2*bound dots, then ||d||^2 and g.d one by one
    assert leg["exchanges_per_two_loop"] == pytest.approx(2 * 5 + 2)
    assert j["roofline"]["exchanges_per_two_loop"] == pytest.approx(12.0)
    assert "exchange_us_mean" in j["roofline"] and "ranks_seen" in leg


def test_the_n_gt_1_line_says_which_leg_its_value_comes_from():
    """BASELINE.json's north star names "a scalar RCCL all-reduce"; bench.py reports the BEST leg as `value`, which on real
    hardware will most likely be the in-kernel P2P exchange.  So the line says so itself: `metric` ends in the leg in words,
    `config.allreduce_says` repeats it, and `config.rccl` carries the RCCL leg's figures -- or why there are none -- at the
    top level of `config`, whichever leg won (round-4 verdict, weak #4)."""
    env = dict(os.environ, LBFGS_BENCH_WORKER=os.path.join(ROOT, "tests", "support", "bench_on_mock.py"), OMP_NUM_THREADS="1",
               LBFGS_BENCH_LEGS="rccl,callback")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "12", "--dim", "3000", "--hist",
           "5", "--repeats", "2", "--no-vector-free", "--no-cpu-baseline", "--probe-timeout", "60"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    cfg = j["config"]
    assert cfg["allreduce"] == cfg["value_from_leg"] == "callback"
    assert j["metric"].startswith(bench.METRIC) and j["metric"].endswith(bench.LEG_SAYS["callback"]) and "NOT RCCL" in j["metric"]
    assert cfg["allreduce_says"] == bench.LEG_SAYS["callback"]
    # the test double has no RCCL: the probe failed, the leg never ran, and the record says exactly that
    r = cfg["rccl"]
    assert r["iters_per_sec"] is None and r["two_loop_ms"] is None and r["allreduce_us_mean"] is None
    assert r["status"] == "not run" and cfg["probes"]["rccl"]["status"] != "ok" and "ncclAllReduce" in r["says"]
    # every leg has its sentence, and only the two rccl ones may be read as RCCL numbers
    for leg, says in bench.LEG_SAYS.items():
        assert ("NOT RCCL" in says) == (leg not in ("rccl", "rccl-per-step", "none")), leg


def test_rccl_beside_reads_the_rccl_leg_whichever_leg_won():
    line = {"value": 512.25, "roofline": {"two_loop": {"ms": 1.5}},
            "config": {"comm_info": {"exchange_us_mean": 21.0, "exchanges_per_two_loop": 21.0, "ranks_seen": 8}}}
    p2p = {"value": 640.0, "roofline": {}, "config": {"comm_info": {}}}
    got = bench.rccl_beside({"p2p": {"status": "ok"}, "rccl": {"status": "ok"}}, [("p2p", p2p), ("rccl", line)])
    assert got["iters_per_sec"] == 512.25 and got["two_loop_ms"] == 1.5 and got["allreduce_us_mean"] == 21.0
    assert got["allreduces_per_two_loop"] == 21.0 and got["ranks_seen"] == 8 and got["status"] == "ok"
    got = bench.rccl_beside({"rccl": {"status": "timed out after 150 s (killed)"}}, [("p2p", p2p)])
    assert got["iters_per_sec"] is None and "timed out" in got["status"]


def test_a_leg_is_named_by_what_ran_not_by_what_was_asked_for():
    """"rccl" asks for the gated exchange under the persistent kernel; lbfgs_hip_ctx_create may decide otherwise for every rank
    (its trial of the exchange failed somewhere) and the run is then RCCL with a kernel per step.  The line must not call that
    "gated": the supervisor files a measurement under the leg its record shows (roofline.two_loop.resident_kernel)."""
    gated = {"value": 900.0, "roofline": {"two_loop": {"ms": 0.8, "resident_kernel": True}}, "config": {"comm_info": {}}}
    fell_back = {"value": 500.0, "roofline": {"two_loop": {"ms": 1.7, "resident_kernel": False}}, "config": {"comm_info": {"ranks_seen": 8}}}
    assert bench.ran_as("rccl", gated) == "rccl" and bench.ran_as("rccl", fell_back) == "rccl-per-step"
    assert bench.ran_as("p2p", fell_back) == "p2p-per-step" and bench.ran_as("p2p", gated) == "p2p"
    for leg in ("p2p-host", "callback", "rccl-per-step", "p2p-per-step"):
        assert bench.ran_as(leg, fell_back) == leg
    assert bench.ran_as("rccl", {"value": 1.0, "config": {}}) == "rccl"   # (no record of the launch form: nothing to correct)
    got = bench.rccl_beside({"rccl": {"status": "ok", "ran_as": "rccl-per-step"}}, [(bench.ran_as("rccl", fell_back), fell_back)])
    assert got["iters_per_sec"] == 500.0 and got["says"] == bench.LEG_SAYS["rccl-per-step"] and "gated" not in got["says"]


def test_world_8_through_the_supervisor_on_the_test_double():
    """The metric's own world through bench.py's N > 1 launch form (plain `python bench.py --gpus 8`: one
    torch.distributed.run child of eight ranks per job) on the CPU test double: eight shards -- seven of 12 500 224 / 1e8-like
    proportion and the short last one --, probes and legs for eight children inside the budget, the CPU baseline attached.
    (Eight rank PROCESSES cannot share one GPU on the pool -- six at most -- so the GPU rehearsal of world 8 hosts two ranks per
    process: tools/eight_ranks_one_gpu.py, profiles/r05_bench_eight_ranks_sharing_one_gpu.json.)"""
    env = dict(os.environ, LBFGS_BENCH_WORKER=os.path.join(ROOT, "tests", "support", "bench_on_mock.py"), OMP_NUM_THREADS="1")
    n = 100_003
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "12", "--dim", str(n), "--hist",
           "5", "--repeats", "2", "--no-vector-free", "--comm", "callback", "--cpu-n", "20000", "--total-budget", "400"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["value"] > 0 and "8 GPUs" in j["metric"]
    lo, hi = shard_range(n, 0, 8)
    assert j["config"]["n_local_rank0"] == hi - lo == 12544        # ceil(n / 8) rounded up to 256 elements
    assert shard_range(n, 7, 8) == (7 * 12544, n) and n - 7 * 12544 < 12544   # ... and the short last shard
    leg = j["config"]["legs"]["callback"]
    assert leg["status"] == "ok" and leg["exchanges_per_two_loop"] == pytest.approx(2 * 5 + 2)
    assert j["config"]["comm_info"]["world"] == 8 and j["config"]["budget"]["used_s"] < 400
    assert j["cpu_baseline"]["value"] > 0 and j["config"]["rccl"]["status"] == "not run"


# ---------------------------------------------------------------------------------------------
# roofline.traffic taken by the run itself: two rocprofv3 --pmc child runs after the timed region (bench.live_traffic)
# ---------------------------------------------------------------------------------------------
FAKE_ROCPROF = r'''#!/usr/bin/env python3
# stands in for rocprofv3 in the CPU suite: checks the command line bench.py builds and lays the committed raw counter rows
# out the way rocprofv3 does (<-d>/<host>/<pid>_counter_collection.csv)
import os, shutil, sys, time
argv = sys.argv[1:]
mode = os.environ.get("FAKE_ROCPROF_MODE", "ok")
stats = "--stats" in argv
if stats:
    assert "--pmc" not in argv, argv            # counters and statistics never share a pass
else:
    assert argv[0] == "--pmc" and argv[1] in ("FETCH_SIZE", "WRITE_SIZE"), argv
assert "--kernel-trace" in argv and "--sys-trace" not in argv and "--hip-trace" not in argv and "-s" not in argv, argv
out = argv[argv.index("-d") + 1]
prog = argv[argv.index("--") + 1:]
assert os.path.basename(prog[0]).startswith("python") and prog[1].endswith("bench.py"), prog   # the program itself follows `--`
for flag in ("--no-live-traffic", "--no-prof", "--no-cpu-baseline", "--no-vector-free"):
    assert flag in prog, (flag, prog)
assert os.environ.get("LBFGS_BENCH_LIVE_TRAFFIC") == "0"
if mode == "fail":
    sys.stderr.write("rocprofv3: no agents found\n"); sys.exit(3)
if mode == "hang":
    time.sleep(60)
os.makedirs(os.path.join(out, "box"), exist_ok=True)
if stats:
    k = "void lh::two_loop_resident_kernel<60, true, true>(lh::ResArgs, lh::RedCtl)"
    rows = ["Kernel_Name,Start_Timestamp,End_Timestamp"]
    t = 1000
    for i in range(40):   # the first m launches of a run are shallower (history not yet full): they must not count
        d = 9_200_000 + 1000 * (i % 3) if i >= 10 else 850_000 * (i + 1)
        rows.append(f'"{k}",{t},{t + d}')
        t += d + 2_000_000
    rows.append('"void lh::stream_kernel<lh::OpCopy<false>, 1, 256u, 2u, 0, 1>(lh::Args)",1,5001')
    if mode != "empty":
        open(os.path.join(out, "box", "123_kernel_trace.csv"), "w").write("\n".join(rows) + "\n")
    sys.exit(0)
src = os.environ["FAKE_ROCPROF_ROWS_" + argv[1]]
if mode != "empty":
    shutil.copy(src, os.path.join(out, "box", "123_counter_collection.csv"))
'''


def _fake_rocprof(tmp_path, monkeypatch, mode="ok"):
    exe = tmp_path / "bin" / "rocprofv3"
    exe.parent.mkdir(exist_ok=True)
    exe.write_text(FAKE_ROCPROF)
    exe.chmod(0o755)
    monkeypatch.setenv("PATH", str(exe.parent) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("FAKE_ROCPROF_MODE", mode)
    monkeypatch.setenv("FAKE_ROCPROF_ROWS_FETCH_SIZE", os.path.join(ROOT, "profiles", "r06_pmc_fetch_counter_collection.csv"))
    monkeypatch.setenv("FAKE_ROCPROF_ROWS_WRITE_SIZE", os.path.join(ROOT, "profiles", "r06_pmc_write_counter_collection.csv"))


def _args(**kw):
    import argparse
    base = dict(n=100_000_000, m=10, warmup=12, line_eval=2, grid=0, no_live_traffic=False, no_prof=False, gpus=1)
    base.update(kw)
    return argparse.Namespace(**base)


def test_live_traffic_reads_the_counter_rows_of_its_own_child_runs(tmp_path, monkeypatch):
    """The orchestration without a GPU: the command lines are what /opt/skills/guides prescribe (one counter per pass,
    --kernel-trace only, the program itself behind `--`), the children cannot recurse, and the figure is the summariser's
    arithmetic over the rows -- here the committed rows of this build, so the live figure must equal the committed one."""
    _fake_rocprof(tmp_path, monkeypatch)
    roof = {"kernel": "two_loop_resident_kernel<ER,NT>", "achieved": 6400.0}
    roof.update(bench.traffic_lookup(100_000_000, 10, roof["kernel"]))
    committed = roof["traffic"]
    live = bench.live_traffic(_args(), roof, timeout_s=30.0)
    assert live["traffic_live"] == pytest.approx(committed, rel=1e-9) and live["traffic_live_kernel"].startswith("two_loop_resident_kernel<60")
    assert live["traffic_live_read_GB"] > 40 and live["traffic_live_write_GB"] > 10
    out = bench.apply_live_traffic(dict(roof), live)
    assert out["traffic"] == live["traffic_live"] and out["traffic_committed"] == committed
    assert out["traffic_is_current"] is True and out["traffic_file"] is None and "this run" in out["traffic_source"]
    assert out["traffic_committed_file"] == "profiles/pmc_traffic.json"


@pytest.mark.parametrize("mode,says", [("fail", "exited with 3"), ("empty", "no FETCH_SIZE rows"), ("hang", "did not finish")])
def test_live_traffic_failures_leave_the_committed_figure_in_place(tmp_path, monkeypatch, mode, says):
    _fake_rocprof(tmp_path, monkeypatch, mode)
    roof = {"kernel": "two_loop_resident_kernel<ER,NT>", "achieved": 6400.0}
    roof.update(bench.traffic_lookup(100_000_000, 10, roof["kernel"]))
    before = dict(roof)
    live = bench.live_traffic(_args(), roof, timeout_s=2.0 if mode == "hang" else 30.0)
    assert live["traffic_live"] is None and says in live["traffic_live_error"]
    out = bench.apply_live_traffic(dict(roof), live)
    assert out["traffic"] == before["traffic"] and out["traffic_file"] == before["traffic_file"] and "traffic_committed" not in out
    assert not [d for d in os.listdir("/tmp") if d.startswith("lbfgs_bench_pmc_") and os.path.getmtime(os.path.join("/tmp", d)) > time.time() - 1.0]


def test_live_traffic_is_for_the_plain_single_gpu_run_only(monkeypatch):
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")) or k in ("RANK", "WORLD_SIZE", "LBFGS_BENCH_LIVE_TRAFFIC", "LBFGS_TEST_BACKEND", "LBFGS_BENCH_WORKER"):
            monkeypatch.delenv(k)
    assert bench.live_traffic_wanted(_args())
    assert not bench.live_traffic_wanted(_args(no_live_traffic=True)) and not bench.live_traffic_wanted(_args(no_prof=True))
    assert not bench.live_traffic_wanted(_args(gpus=8))
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "2"), ("ROCPROFILER_LIBRARY_CTOR", "1"), ("ROCP_TOOL_LIBRARIES", "x"),
                 ("LBFGS_BENCH_LIVE_TRAFFIC", "0"), ("LBFGS_TEST_BACKEND", "mock"), ("LBFGS_BENCH_WORKER", "x")):
        monkeypatch.setenv(k, v)
        assert not bench.live_traffic_wanted(_args()), k
        monkeypatch.delenv(k)


def test_live_kernel_time_prices_the_full_depth_dispatches(tmp_path, monkeypatch):
    """roofline.rocprofv3_avg_ms: a `--kernel-trace --stats` child run of the same command, the full-depth dispatches only (the
    first m two-loops of a run are shallower), beside the HIP-event figure; no counters in that pass."""
    _fake_rocprof(tmp_path, monkeypatch)
    roof = {"kernel": "two_loop_resident_kernel<ER,NT>", "avg_ms": 9.3, "bytes_per_launch": 59972483072}
    got = bench.live_kernel_time(_args(), roof, timeout_s=30.0)
    assert got["rocprofv3_launches"] == 30 and got["rocprofv3_avg_ms"] == pytest.approx(9.201, abs=2e-3)
    assert got["rocprofv3_over_hip_events"] == pytest.approx(9.201 / 9.3, abs=1e-3)
    assert got["frac_on_rocprofv3_time"] == pytest.approx(59.972483072 / 9.201e-3 / 8000.0, rel=1e-3)
    monkeypatch.setenv("FAKE_ROCPROF_MODE", "empty")
    got = bench.live_kernel_time(_args(), roof, timeout_s=30.0)
    assert got["rocprofv3_avg_ms"] is None and "no dispatches" in got["rocprofv3_error"]
    monkeypatch.setenv("FAKE_ROCPROF_MODE", "fail")
    got = bench.live_kernel_time(_args(), roof, timeout_s=30.0)
    assert got["rocprofv3_avg_ms"] is None and "exited with 3" in got["rocprofv3_error"]


# ---------------------------------------------------------------------------------------------
# N = 1: the line exists as soon as the measurement returns; what follows runs inside ONE budget; a signal prints the line so far
# (round-5 verdict, weak #4: the only JSON line used to be written after up to 3 x 150 s of rocprofv3 child runs and a 420 s wait)
# ---------------------------------------------------------------------------------------------
def _n1_on_the_test_double(tmp_path, mode, extra_env=None, extra_args=()):
    exe = tmp_path / "bin" / "rocprofv3"
    exe.parent.mkdir(exist_ok=True)
    exe.write_text(FAKE_ROCPROF)
    exe.chmod(0o755)
    env = dict(os.environ, OMP_NUM_THREADS="1", LBFGS_BENCH_LIVE_TRAFFIC="force", FAKE_ROCPROF_MODE=mode, LBFGS_MOCK_FAKE_KERNEL_TIMES="1",
               PATH=str(exe.parent) + os.pathsep + os.environ["PATH"],
               FAKE_ROCPROF_ROWS_FETCH_SIZE=os.path.join(ROOT, "profiles", "r06_pmc_fetch_counter_collection.csv"),
               FAKE_ROCPROF_ROWS_WRITE_SIZE=os.path.join(ROOT, "profiles", "r06_pmc_write_counter_collection.csv"))
    for k in list(env):
        if k.startswith(("ROCPROF", "ROCP_")) or k in ("RANK", "WORLD_SIZE"):
            del env[k]
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, "tests", "support", "bench_on_mock.py"), "--steps", "4", "--warmup", "12", "--dim", "3000",
           "--hist", "5", "--repeats", "2", "--cpu-n", "3000", "--no-vector-free"] + list(extra_args)
    return subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _one_line(out):
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_n1_line_appears_within_budget_when_every_rocprofv3_child_hangs(tmp_path):
    """Three stand-in rocprofv3 runs that never return, each bounded by --rocprof-timeout, all of them inside --post-budget: the
    third is not even started once less than its timeout is left, and the line says what was cut."""
    t0 = time.monotonic()
    p = _n1_on_the_test_double(tmp_path, "hang", extra_args=["--post-budget", "9", "--rocprof-timeout", "3"])
    out, err = p.communicate(timeout=120)
    took = time.monotonic() - t0
    assert p.returncode == 0, err[-3000:]
    j = _one_line(out)
    assert j["value"] > 0 and j["cpu_baseline"]["value"] > 0
    roof = j["roofline"]
    assert "did not finish within 3 s" in roof["traffic_live_error"] and roof.get("traffic_live") is None
    assert "did not finish within 3 s" in roof["rocprofv3_error"] or "skipped" in roof["rocprofv3_error"]
    cut = j["config"]["budget"]["cut"]
    assert any(c.startswith("traffic_live_error") for c in cut) and any(c.startswith("rocprofv3_error") for c in cut)
    assert j["config"]["budget"]["used_s"] < 9 + 10 and took < 60


def test_n1_a_pass_is_skipped_when_less_than_its_timeout_is_left(tmp_path):
    p = _n1_on_the_test_double(tmp_path, "ok", extra_args=["--post-budget", "5", "--rocprof-timeout", "30"])
    out, err = p.communicate(timeout=120)
    assert p.returncode == 0, err[-3000:]
    roof = _one_line(out)["roofline"]
    assert roof["traffic_live_error"].startswith("the FETCH_SIZE pass skipped:") and "--post-budget" in roof["traffic_live_error"]
    assert roof["rocprofv3_error"].startswith("the --kernel-trace --stats pass skipped:")


def test_n1_line_appears_within_budget_when_the_baseline_child_never_returns(tmp_path):
    t0 = time.monotonic()
    p = _n1_on_the_test_double(tmp_path, "ok", extra_env={"LBFGS_BENCH_TEST_CPU_CHILD": "hang"}, extra_args=["--post-budget", "8"])
    out, err = p.communicate(timeout=120)
    took = time.monotonic() - t0
    assert p.returncode == 0, err[-3000:]
    j = _one_line(out)
    assert j["value"] > 0 and j["cpu_baseline"]["value"] is None and j["cpu_baseline"]["sample"].startswith("cut:")
    assert j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["kind"] == "port"
    assert any(c.startswith("cpu_baseline: cut") for c in j["config"]["budget"]["cut"])
    assert took < 60
    # ... and the child is gone (it would hold ~22 GB of host memory at the metric's size)
    left_over = subprocess.run(["pgrep", "-f", "time.sleep(100000)"], capture_output=True, text=True).stdout.split()
    assert not left_over, left_over


@pytest.mark.parametrize("sig", ["SIGTERM", "SIGINT"])
def test_n1_a_signal_prints_the_line_measured_so_far(tmp_path, sig):
    """The driver's kill while a rocprofv3 child hangs and the baseline child never returns: the measured line is printed at
    once, says what it lacks, and neither child survives."""
    import signal as _signal

    p = _n1_on_the_test_double(tmp_path, "hang", extra_env={"LBFGS_BENCH_TEST_CPU_CHILD": "hang"},
                               extra_args=["--post-budget", "200", "--rocprof-timeout", "60"])
    # the stand-in rocprofv3 announces itself on stderr through bench.py's own progress line
    deadline = time.monotonic() + 90
    seen = ""
    os.set_blocking(p.stderr.fileno(), False)
    while time.monotonic() < deadline and "rocprofv3 --pmc FETCH_SIZE pass" not in seen:
        try:
            seen += p.stderr.read() or ""
        except (BlockingIOError, TypeError):
            pass
        time.sleep(0.2)
    assert "rocprofv3 --pmc FETCH_SIZE pass" in seen, seen[-2000:]
    time.sleep(0.5)
    t0 = time.monotonic()
    p.send_signal(getattr(_signal, sig))
    out, _ = p.communicate(timeout=60)
    assert time.monotonic() - t0 < 20 and p.returncode == 0
    j = _one_line(out)
    assert j["value"] > 0 and j["config"]["interrupted_by_signal"] == int(getattr(_signal, sig))
    assert j["cpu_baseline"]["value"] is None and "signal" in j["cpu_baseline"]["sample"]
    assert j["roofline"].get("traffic_live") is None
    time.sleep(0.5)
    left_over = subprocess.run(["pgrep", "-f", "time.sleep(100000)"], capture_output=True, text=True).stdout.split()
    assert not left_over, left_over


# ---------------------------------------------------------------------------------------------
# Round 6: first contact with several ranks, blind (round-5 verdict, next #3).  The rank processes run on the test double, whose
# P2P and RCCL stand-ins live in POSIX shared memory (tests/support/mock_lbfgs_hip.cpp): the legs really exchange.
# ---------------------------------------------------------------------------------------------
def test_a_rank_that_sees_only_its_own_device_takes_it():
    assert bench.pick_device(5, -1, 8) == 5          # one process per GPU, all devices visible: LOCAL_RANK
    assert bench.pick_device(5, -1, 1) == 0          # masked per rank: the one device this rank sees
    assert bench.pick_device(5, -1, 2) == 1 and bench.pick_device(5, 3, 1) == 3 and bench.pick_device(2, -1, 0) == 2


def _supervised(args, env_extra, timeout=600):
    env = dict(os.environ, LBFGS_BENCH_WORKER=os.path.join(ROOT, "tests", "support", "bench_on_mock.py"), OMP_NUM_THREADS="1",
               LBFGS_MOCK_FAKE_KERNEL_TIMES="1")
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_world_8_with_devices_masked_per_rank_ends_on_host_placed_mailboxes():
    """`bench.py --gpus 8` where every rank can map only its OWN device's memory (a launcher that masks devices per rank, or a
    machine without peer access): the "p2p" probe (device mailboxes, strictly) fails on every rank TOGETHER with the IPC error,
    nothing hangs, and the run goes on with the legs that do work -- host-placed mailboxes and RCCL -- and says which one
    `value` comes from."""
    j, err = _supervised(["--gpus", "8", "--steps", "4", "--warmup", "12", "--dim", "100003", "--hist", "5", "--repeats", "2",
                          "--no-vector-free", "--no-cpu-baseline", "--total-budget", "400", "--pg-backend", "gloo"],
                         {"LBFGS_MOCK_NO_DEVICE_IPC": "1", "LBFGS_MOCK_RCCL": "1"})
    cfg = j["config"]
    assert cfg["probes"]["p2p"]["status"] != "ok" and cfg["probes"]["p2p-host"]["status"] == "ok" and cfg["probes"]["rccl"]["status"] == "ok"
    assert "hipIpcOpenMemHandle" in err
    assert "p2p" not in cfg["legs"] and cfg["legs"]["p2p-host"]["status"] == "ok" and cfg["legs"]["rccl"]["status"] == "ok"
    host = cfg["legs"]["p2p-host"]
    assert host["mailbox_placement"] == "host" and host["ranks_seen"] == 8 and host["mailboxes_mapped"] == [0, 7]
    assert cfg["legs"]["rccl"]["ranks_seen"] == 8
    assert cfg["value_from_leg"] in ("p2p-host", "rccl") and j["n_gpus"] == 8 and j["value"] > 0
    assert cfg["rccl"]["iters_per_sec"] > 0 and cfg["budget"]["used_s"] < 400
    for leg in ("p2p-host", "rccl"):   # the per-leg figures one real run will be read by (all there, None where the double has none)
        for k in ("exchange_us_mean", "exchange_us_p50", "exchange_us_p99", "exchange_us_max", "local_wait_us_mean", "local_wait_us_max",
                  "exchanges_per_two_loop", "two_loop_ms"):
            assert k in cfg["legs"][leg], (leg, k)


@pytest.mark.parametrize("fault", [False, True], ids=["gated_agreed", "trial_fails_on_one_rank"])
def test_the_rccl_leg_is_filed_by_what_every_rank_agreed_on(fault):
    """bench.py's "rccl" leg OPTS IN to the gated exchange (LBFGS_HIP_RCCL_RESIDENT=1: the library's default is a kernel per
    step).  If the trial at context creation fails on exactly ONE rank after the handshake passed everywhere, every rank must
    take the kernel-per-step form (the product's collective skeleton, here on three processes of the test double) and the line
    must call the measurement what it was: "rccl-per-step" -- in `metric`, `config.allreduce`, `config.rccl.says`."""
    env = {"LBFGS_MOCK_RCCL": "1", "LBFGS_BENCH_LEGS": "rccl"}
    if fault:
        env.update(LBFGS_HIP_RESIDENT_FAULT="-1", LBFGS_MOCK_FAULT_RANK="1")
    j, err = _supervised(["--gpus", "3", "--steps", "4", "--warmup", "12", "--dim", "5000", "--hist", "5", "--repeats", "2",
                          "--no-vector-free", "--no-cpu-baseline", "--probe-timeout", "60", "--pg-backend", "gloo"], env)
    cfg = j["config"]
    leg = cfg["legs"]["rccl"]
    assert leg["status"] == "ok" and leg["ranks_seen"] == 3
    want = "rccl-per-step" if fault else "rccl"
    assert cfg["allreduce"] == cfg["value_from_leg"] == want and j["metric"].endswith(bench.LEG_SAYS[want])
    assert cfg["rccl"]["says"] == bench.LEG_SAYS[want] and cfg["rccl"]["iters_per_sec"] == pytest.approx(j["value"], rel=1e-3)
    assert (leg.get("ran_as") == "rccl-per-step") == fault
    assert j["roofline"]["two_loop"]["resident_kernel"] is (not fault)
    assert ("gated RCCL exchange is not used" in err) == fault


def test_exchange_quantiles_from_the_device_histogram():
    """lbfgs_hip_comm_info.exchange_hist (ABI 5): quarter-microsecond bins up to 8 us, octaves beyond; bench.py's p50 / p99."""
    from rust_lbfgs_amd.api import exchange_bin_edges_us, exchange_quantile

    edges = exchange_bin_edges_us()
    assert len(edges) == 48 and edges[0] == (0.0, 0.25) and edges[31] == (7.75, 8.0) and edges[32] == (8.0, 16.0) and edges[33] == (16.0, 32.0)
    assert edges[-1][1] == float("inf")
    h = [0] * 48
    assert exchange_quantile(h, 0.5) is None
    h[16] = 98          # 4.00 .. 4.25 us
    h[34] = 2           # 32 .. 64 us: two stragglers in a hundred
    assert 4.0 <= exchange_quantile(h, 0.50) <= 4.25 and 4.0 <= exchange_quantile(h, 0.98) <= 4.25
    assert 32.0 <= exchange_quantile(h, 0.99) <= 64.0 and exchange_quantile(h, 1.0) == 64.0
    h = [0] * 48
    h[47] = 5
    assert exchange_quantile(h, 0.5) == edges[47][0]
