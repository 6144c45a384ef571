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
