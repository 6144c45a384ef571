#!/usr/bin/env python3
"""The step-locked test (tests/test_gpu_step_locked.py: the GPU is handed the oracle's inputs of every iteration and must
reproduce its outputs within 1e-10) on the configuration of a seed of the random sweep (Rosenbrock seeds only: the sweep's
other objectives start from a drawn x0 the step-locked cases do not take).  For reading outliers of tools/fuzz_soak.py.
Mind what it can and cannot say: at an iteration whose newest pair has y == g (a step so large that gp drops below half an ulp
of g) the oracle's alpha is EXACTLY -1 because ys and s.q are the same sum; handing the GPU the oracle's ys breaks that tie, so
`d` may read far off there although every sum is within 1e-14 (seed 28675).
    python tools/steplock_fuzz.py 28675 79105"""
import sys, traceback
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from rust_lbfgs_amd import objectives
from tests import fuzz_common as F
import tests.test_gpu_step_locked as S
for seed in [int(s) for s in sys.argv[1:]]:
    c = F.make_case(seed)
    assert c["kind"] == "rosenbrock"
    S.CASES["fuzz"] = (c["n"], c["m"], (lambda cc: (lambda b: F.configure(b, cc)))(c), O.rosenbrock, objectives.Rosenbrock, "rosenbrock", c["iters"])
    print("seed", seed, c, flush=True)
    try:
        S.test_step_locked("fuzz")
        print("  step-locked: within 1e-10 at every iteration")
    except AssertionError:
        print("  ", traceback.format_exc().splitlines()[-1][:500])
