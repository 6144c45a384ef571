# experiment: which (big, small) grid pairs of two processes' resident kernels are co-resident on one GPU?
import json, os, sys, time, pathlib, tempfile
sys.path.insert(0, os.getcwd())
os.environ["LBFGS_WORKER_PRODUCT"] = "1"
os.environ["LBFGS_COMM_KIND"] = "p2p"
os.environ["LBFGS_TEST_EXCLUSIVE_DEVICE"] = "1"
os.environ["LBFGS_HIP_HANDOFF_TIMEOUT_MS"] = "1500"
from tests.test_distributed_cpu import run_world
n, cut = 3_000_000 + 40_960 + 5, 3_000_000
os.environ["LBFGS_TEST_BOUNDS"] = json.dumps([0, cut, n])
for grids in sys.argv[1:]:
    os.environ["LBFGS_TEST_RESIDENT_GRIDS"] = grids
    t0 = time.time()
    with tempfile.TemporaryDirectory() as d:
        try:
            outs = run_world(dict(name="probe", n=n, m=6, iters=8, objective="quadratic"), 2, pathlib.Path(d))
            print(grids, "->", [(o["err"], o["errmsg"][:60], o["resident"]) for o in outs], f"{time.time()-t0:.1f} s", flush=True)
        except AssertionError as e:
            print(grids, "-> worker failed", str(e)[-300:], flush=True)
