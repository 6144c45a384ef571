#!/usr/bin/env python3
"""Soak of the sharded random sweep: two processes on one GPU, in-kernel P2P exchange (or LBFGS_COMM_KIND=callback).
    python tools/fuzz_soak_p2p.py 0 40"""
import os
import pathlib
import sys
import tempfile
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LBFGS_WORKER_PRODUCT"] = "1"
os.environ.setdefault("LBFGS_COMM_KIND", "p2p")
from tests.test_distributed_cpu import compare_sharded_fuzz  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    with tempfile.TemporaryDirectory() as d:
        try:
            compare_sharded_fuzz(seed, pathlib.Path(d), vector_free=bool(seed % 2))
        except Exception:  # noqa: BLE001
            bad.append(seed)
            print("FAIL seed", seed, traceback.format_exc().splitlines()[-1][:500], flush=True)
    if seed % 10 == 0:
        print("... seed", seed, "failures so far", len(bad), flush=True)
print("seeds", lo, "..", hi - 1, "failures:", bad)
sys.exit(1 if bad else 0)
