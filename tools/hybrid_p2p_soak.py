#!/usr/bin/env python3
"""Soak of the HYBRID persistent two-loop kernel under the in-kernel P2P exchange: two processes on one GPU, one to three
workgroups each (so that a few 1e5 elements per rank already exceed "the chip"), random sizes / history lengths / objectives,
whole runs against the single-rank oracle.      python tools/hybrid_p2p_soak.py 0 12"""
import os
import pathlib
import sys
import tempfile
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LBFGS_WORKER_PRODUCT"] = "1"
os.environ.setdefault("LBFGS_COMM_KIND", "p2p")   # (LBFGS_COMM_KIND=p2p-host: mailboxes in host shared memory)
os.environ["LBFGS_TEST_EXCLUSIVE_DEVICE"] = "1"
from tests.test_distributed_cpu import oracle_rows, run_world  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    r = np.random.default_rng(1000 + seed)
    grid = int(r.integers(1, 4))
    n = int(r.integers(2 * 96 * 256 * grid * 2 + 1000, 2 * 96 * 256 * grid * 2 * 3))  # each of the 2 ranks: 1x .. 3x "the chip"
    case = dict(name=f"hyb{seed}", n=n, m=int(r.integers(1, 9)), iters=int(r.integers(6, 16)),
                objective="logistic" if seed % 3 == 2 else "quadratic")
    if seed % 3 == 2:
        case["owl"] = [0.5, int(n * 0.1), int(n * 0.9)]
    os.environ["LBFGS_HIP_RESIDENT_GRID"] = str(grid)
    with tempfile.TemporaryDirectory() as d:
        try:
            outs = run_world(case, 2, pathlib.Path(d))
            ref_rows, ref_x = oracle_rows(case)
            assert outs[0]["rows"] == outs[1]["rows"] and len(outs[0]["rows"]) == len(ref_rows)
            for o in outs:
                assert o["resident"] >= 1 and 0 < o["resident_elements"] < o["hi"] - o["lo"], (o["resident"], o["resident_elements"])
            for got, ref in zip(outs[0]["rows"], ref_rows):
                assert got[:3] == ref[:3]
                for a, b in zip(got[3:], ref[3:]):
                    assert abs(a - b) <= 1e-9 * max(abs(b), 1e-6), (got, ref)
            x = np.concatenate([np.array(o["x"]) for o in outs])
            assert np.max(np.abs(x - ref_x)) <= 1e-9 * max(np.max(np.abs(ref_x)), 1e-12)
            print("ok  ", case, "grid", grid, flush=True)
        except Exception:  # noqa: BLE001
            bad.append(seed)
            print("FAIL", case, "grid", grid, traceback.format_exc().splitlines()[-1][:300], flush=True)
print("seeds", lo, "..", hi - 1, "failures:", bad)
sys.exit(1 if bad else 0)
