"""FIRST CONTACT with more than one GPU.  Everything in this file needs at least two devices and is SKIPPED on the one-GPU boxes
this build has been developed on (six rounds, no multi-GPU node): it is the list of things that have never executed -- RCCL with
peers, HIP IPC across devices, stores over xGMI, the gated exchange with a real ncclAllReduce under the persistent kernel -- as
tests that run the moment a node offers two devices, one rank per device, against the single-rank CPU oracle.

(Round-5 advice: "a 2-rank, 2-GPU test with uneven bounds, including an empty shard, that compares the gated trajectory with
the per-step one".)"""
import json
import os

import numpy as np
import pytest

from tests.test_distributed_cpu import oracle_rows, run_world

pytestmark = pytest.mark.gpu


def _devices():
    try:
        import torch

        return int(torch.cuda.device_count())   # (counting devices does not initialise the GPU)
    except Exception:  # noqa: BLE001
        return 0


NDEV = _devices()
needs2 = pytest.mark.skipif(NDEV < 2, reason=f"needs >= 2 GPUs (this box has {NDEV})")
needs3 = pytest.mark.skipif(NDEV < 3, reason=f"needs >= 3 GPUs (this box has {NDEV})")


def _check(case, outs, tol=1e-9):
    ref_rows, ref_x = oracle_rows(case)
    assert all(o["err"] == 0 for o in outs), [o["errmsg"] for o in outs]
    for o in outs[1:]:
        assert o["rows"] == outs[0]["rows"]            # every rank saw the same global scalars, bit for bit
    assert len(outs[0]["rows"]) == len(ref_rows)
    for got, ref in zip(outs[0]["rows"], ref_rows):
        assert got[:3] == ref[:3]
        for a, b in zip(got[3:], ref[3:]):
            assert abs(a - b) <= tol * max(abs(b), 1e-6), (got, ref)
    x = np.concatenate([np.array(o["x"])[: o["hi"] - o["lo"]] for o in outs])
    assert np.max(np.abs(x - ref_x)) <= tol * max(np.max(np.abs(ref_x)), 1e-12)


def _env(monkeypatch, kind, **extra):
    monkeypatch.setenv("LBFGS_WORKER_PRODUCT", "1")
    monkeypatch.setenv("LBFGS_TEST_DEVICE_PER_RANK", "1")
    monkeypatch.setenv("LBFGS_TEST_EXCLUSIVE_DEVICE", "1")
    monkeypatch.setenv("LBFGS_COMM_KIND", kind)
    for k, v in extra.items():
        monkeypatch.setenv(k, v)


@needs2
@pytest.mark.parametrize("kind", ["p2p-device", "p2p-host", "p2p"])
@pytest.mark.parametrize("case", [
    dict(name="on_chip", n=2_600_001, m=6, iters=14, objective="quadratic"),
    dict(name="hybrid", n=30_000_003, m=5, iters=9, objective="quadratic"),
    dict(name="owlqn", n=900_001, m=6, iters=12, objective="logistic", owl=[0.5, 300_000, 850_000]),
], ids=lambda c: c["name"] if isinstance(c, dict) else c)
def test_two_gpus_p2p_exchange_inside_the_kernels(case, kind, tmp_path, monkeypatch):
    """Mailboxes in the peer's DEVICE memory mapped through HIP IPC and reached over xGMI (never done: one GPU can only map its
    own), host-placed mailboxes, and the collective fall-back between the two; the two-loop as the persistent kernel."""
    _env(monkeypatch, kind)
    outs = run_world(case, 2, tmp_path)
    _check(case, outs)
    assert all(o["resident"] > 0 for o in outs)
    if kind != "p2p":
        assert all(o["placement"] == {"p2p-device": "device", "p2p-host": "host"}[kind] for o in outs)


@needs2
@pytest.mark.parametrize("bounds", [None, "uneven", "empty_last"])
def test_two_gpus_rccl_gated_equals_per_step_equals_oracle(bounds, tmp_path, monkeypatch):
    """RCCL with a real peer.  The gated exchange (opt-in) keeps the persistent kernel and serves its hand-offs with
    ncclAllReduce launches behind gate kernels on a second stream; it must give the per-step form's trajectory and the oracle's.
    With an EMPTY shard the ranks must agree at context creation that nobody takes the gated form (resident == 0 on both)."""
    n = 3_000_005
    case = dict(name="rccl", n=n, m=6, iters=12, objective="quadratic")
    b = {None: None, "uneven": [0, 2_000_128, n], "empty_last": [0, n, n]}[bounds]
    runs = {}
    for form, flag in (("gated", "1"), ("per_step", "0")):
        _env(monkeypatch, "rccl", LBFGS_HIP_RCCL_RESIDENT=flag)
        if b is not None:
            monkeypatch.setenv("LBFGS_TEST_BOUNDS", json.dumps(b))
        d = tmp_path / form
        d.mkdir()
        runs[form] = run_world(case, 2, d)
        _check(case, runs[form])
    gated, per = runs["gated"], runs["per_step"]
    assert all(o["resident"] == 0 for o in per)
    if bounds == "empty_last":
        assert all(o["resident"] == 0 for o in gated)          # agreed: an empty shard cannot take the persistent kernel
    else:
        assert all(o["resident"] > 0 for o in gated)           # (if the trial at context creation failed, stderr says so: read it)
    for a, c in zip(gated[0]["rows"], per[0]["rows"]):
        assert a[:3] == c[:3]
        for u, v in zip(a[3:], c[3:]):
            assert abs(u - v) <= 1e-10 * max(abs(v), 1e-6)


@needs3
def test_three_gpus_with_an_empty_shard_in_the_middle_of_nowhere(tmp_path, monkeypatch):
    """three ranks, the last one empty, P2P: the reductions still close on every rank (an empty rank contributes zeros)"""
    n = 1_000_003
    case = dict(name="q3", n=n, m=5, iters=10, objective="quadratic")
    _env(monkeypatch, "p2p", LBFGS_TEST_BOUNDS=json.dumps([0, 600_064, n, n]))
    _check(case, run_world(case, 3, tmp_path))


@needs2
def test_bench_on_two_gpus_prints_one_attributable_line(tmp_path):
    """`python bench.py --gpus 2` at a reduced size: every leg that can work on this node measured, the line names the leg its
    value comes from and carries per leg what an exchange cost (mean, p50, p99, max) -- the figures profiles/r06_scaling_model.md
    is read against."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dim", "25000448", "--steps", "30", "--repeats", "3",
                        "--no-vector-free", "--cpu-n", "2000000", "--no-cpu-full"], cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0
    legs = j["config"]["legs"]
    assert any(v["status"] == "ok" for v in legs.values()), legs
    for name, leg in legs.items():
        if leg["status"] == "ok" and name.startswith(("p2p", "rccl")):
            assert leg["ranks_seen"] == 2 and leg["exchanges_per_two_loop"] >= 10, (name, leg)
            if name.startswith("p2p") or name == "rccl":
                assert 0.0 < leg["exchange_us_p50"] <= leg["exchange_us_p99"], (name, leg)
