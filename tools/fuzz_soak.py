#!/usr/bin/env python3
"""One-off soak of the seeded random sweep on the HIP path with MANY seeds (the test suite runs seeds 0..59):
    python tools/fuzz_soak.py 60 1500
    python tools/fuzz_soak.py --seeds 527,3119,5311      (named seeds, each under BOTH launch forms of the two-loop)
Prints one line per failing seed and a summary; exit code 1 if any seed fails."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_gpu_parity as T  # noqa: E402

if sys.argv[1] == "--seeds":
    seeds = [int(t) for t in sys.argv[2].split(",")]
    todo = [(sd, pth) for sd in seeds for pth in ("resident", "per_step")]
    lo, hi = min(seeds), max(seeds) + 1
else:
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    # both launch forms of the two-loop in turn (tests/test_gpu_parity.py two_loop_path): one resident kernel / a kernel per step
    todo = [(sd, "resident" if sd % 2 == 0 else "per_step") for sd in range(lo, hi)]
bad = []
for seed, path in todo:
    os.environ["LBFGS_HIP_RESIDENT"] = "1" if path == "resident" else "0"
    try:
        T.test_random_configurations_match_oracle(seed, path)
    except Exception:  # noqa: BLE001
        bad.append(seed)
        print("FAIL seed", seed, traceback.format_exc().splitlines()[-1][:600], flush=True)
    if seed % 100 == 0:
        print("... seed", seed, "failures so far", len(bad), flush=True)
    if seed % 500 == 499:
        from tests import fuzz_common as _F

        print("... bar used so far:", {k: _F.COVERAGE[k] for k in ("cases", "truncated", "rows", "compared", "worst_over_flat_bar",
                                                                     "worst_over_floor", "rows_on_calibrated_bar")}, flush=True)
from tests import fuzz_common as F  # noqa: E402

print("seeds", lo, "..", hi - 1, "failures:", len(bad), bad[:40])
print("coverage and how much of the bar was used (exact path):", F.COVERAGE)
sys.exit(1 if bad else 0)
