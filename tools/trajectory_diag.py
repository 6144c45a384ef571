#!/usr/bin/env python3
"""Side by side, iteration by iteration: the oracle and the HIP path, EACH ON ITS OWN TRAJECTORY, for one seed of the random
sweep (tests/fuzz_common.py) -- the scalars of the recursion (ys, alpha), whether y equals g bit for bit (a step so large that
gp is below half an ulp of g), and how far x, g, d and the newest (s, y) are apart.  For reading outliers of tools/fuzz_soak.py.
    python tools/trajectory_diag.py 28675"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rust_lbfgs_amd import objectives  # noqa: E402
from tests import fuzz_common as F  # noqa: E402


def rel(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / nb) if nb > 0 else float(np.linalg.norm(a))


seed = int(sys.argv[1])
c = F.make_case(seed)
print(c)
n, m = c["n"], c["m"]
dev = {"quadratic": objectives.Quadratic, "logistic": objectives.Logistic, "rosenbrock": objectives.Rosenbrock}[c["kind"]](
    fuse_line_eval=c.get("fuse", 2))
so = F.configure(O.lbfgs().with_m(m), c).build(F.x0_of(c), F.oracle_objective(c))
sg = F.configure(R.lbfgs().with_m(m), c).build(F.x0_of(c), dev)
for it in range(c["iters"]):
    eb = so.end
    try:
        po = so.propagate()
    except O.OracleError as e:
        print("oracle:", e)
        break
    try:
        pg = sg.propagate()
    except R.LbfgsError as e:
        print("device:", e)
        break
    ys_g, al_g = sg.history_scalars()
    ys_o, al_o = [so.ys(j) for j in range(m)], [so.alpha(j) for j in range(m)]
    yo, go = so.hist(eb, "y"), so.vec("gx")
    yg, gg = sg.download(f"y{eb}"), sg.download("gx")
    print(f"it {it}: oracle (niter {po['niter']}, ncall {po['ncall']}) fx {po['fx']!r} step {po['step']!r} | device (ncall {pg.ncall}) fx {pg.fx!r} step {pg.step!r}")
    print(f"    x {rel(sg.download('x'), so.vec('x')):.1e}  g {rel(gg, go):.1e}  d {rel(sg.download('d'), so.vec('d')):.1e}  s_new {rel(sg.download(f's{eb}'), so.hist(eb, 's')):.1e}"
          f"  y_new {rel(yg, yo):.1e}   y == g bitwise: oracle {int(np.sum(yo == go))}/{n}, device {int(np.sum(yg == gg))}/{n}")
    print("    ys     oracle " + " ".join(f"{v!r}" for v in ys_o) + "\n           device " + " ".join(f"{v!r}" for v in ys_g))
    print("    alpha  oracle " + " ".join(f"{v!r}" for v in al_o) + "\n           device " + " ".join(f"{v!r}" for v in al_g))
