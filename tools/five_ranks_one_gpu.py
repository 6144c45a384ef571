#!/usr/bin/env python3
"""The widest multi-rank rehearsal ONE GPU allows (the pool admits six GPU processes, the parent counts as one): five ranks, the persistent two-loop
kernel with the P2P exchange inside its hand-offs (32 workgroups each), whole runs against the single-rank oracle -- for both
mailbox placements, plain and OWL-QN, shards that fit "the chip" of 32 workgroups and shards that do not (hybrid).
    python tools/five_ranks_one_gpu.py"""
import json
import os
import pathlib
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LBFGS_WORKER_PRODUCT"] = "1"
os.environ["LBFGS_TEST_EXCLUSIVE_DEVICE"] = "1"
os.environ["LBFGS_HIP_RESIDENT_GRID"] = "32"
from tests.test_distributed_cpu import oracle_rows, run_world  # noqa: E402

bad = 0
for kind in ("p2p", "p2p-host"):
    os.environ["LBFGS_COMM_KIND"] = kind
    for case in (dict(name="q5", n=5 * 1_500_000 + 77, m=6, iters=12, objective="quadratic"),             # on chip (1.57e6 per 32 workgroups)
                 dict(name="q5h", n=5 * 2_600_000 + 5, m=5, iters=10, objective="quadratic"),              # hybrid, 21 MB shards: `nt`
                 dict(name="owl5", n=5 * 400_000 + 1, m=6, iters=12, objective="logistic", owl=[0.5, 300_000, 1_900_000])):
        with tempfile.TemporaryDirectory() as d:
            outs = run_world(case, 5, pathlib.Path(d))
        ref_rows, ref_x = oracle_rows(case)
        ok = all(o["err"] == 0 and o["rows"] == outs[0]["rows"] and o["resident"] >= case["iters"] - 3 for o in outs) and len(outs[0]["rows"]) == len(ref_rows)
        worst = 0.0
        for got, ref in zip(outs[0]["rows"], ref_rows):
            ok = ok and got[:3] == ref[:3]
            worst = max(worst, max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got[3:], ref[3:])))
        x = np.concatenate([np.array(o["x"]) for o in outs])
        xerr = float(np.max(np.abs(x - ref_x)) / max(np.max(np.abs(ref_x)), 1e-12))
        ok = ok and worst <= 1e-9 and xerr <= 1e-9
        bad += not ok
        print(f"{kind:9s} {case['name']:5s} n={case['n']:>9} five ranks: {'ok' if ok else 'FAILED'}; worst scalar deviation {worst:.2e}, x {xerr:.2e}; "
              f"resident launches per rank {[o['resident'] for o in outs]}, on chip {[o['resident_elements'] for o in outs][:2]}.. of {outs[0]['hi'] - outs[0]['lo']}; "
              f"errors {[o['errmsg'][:40] for o in outs if o['err']]}", flush=True)
sys.exit(1 if bad else 0)
