"""Seeded random configurations shared by the CPU (exact) and GPU (tolerance) sweeps."""
import numpy as np

from oracle import oracle as O
from tests import problems as P

ALGOS = ["MoreThuente", "BacktrackingArmijo", "BacktrackingWolfe", "BacktrackingStrongWolfe"]


def make_case(seed):
    r = np.random.default_rng(seed)
    kind = ["quadratic", "logistic", "rosenbrock"][int(r.integers(0, 3))]
    n = int(r.integers(1, 3000))
    if kind == "rosenbrock":
        n = max(2, n - n % 2)
    c = dict(seed=seed, kind=kind, n=n, m=int(r.integers(1, 11)), algo=ALGOS[int(r.integers(0, 4))],
             damping=bool(r.random() < 0.3), iters=int(r.integers(4, 16)),
             max_step=float([1.0, 1e20, 0.05][int(r.integers(0, 3))]), h0=float([1.0, 0.1, 7.0][int(r.integers(0, 3))]),
             gtol=float([0.9, 0.1][int(r.integers(0, 2))]), max_ls=int([20, 3, 2][int(r.integers(0, 3))]),
             gradient_only=bool(r.random() < 0.1), vector_free=False)
    c["owl"] = None
    if r.random() < 0.35:
        start = int(r.integers(0, n))
        end = None if r.random() < 0.4 else int(r.integers(0, n + 20))  # may be <= start (panic) or > n (clamped)
        c["owl"] = (float([1.0, 0.3, 0.0][int(r.integers(0, 3))]), start, end)
    c["x0_scale"] = float([0.0, 0.5, 3.0][int(r.integers(0, 3))])
    # derived from the seed, not drawn, so the cases above keep their values:
    c["fuse"] = [2, 1, 0][seed % 3]          # lbfgs_evaluator.fuse_line_eval: deferred trial points / fused / separate
    if seed % 11 == 5:
        c["max_ls"] = seed % 2               # 0 or 1: the search runs NO trial (line.rs:258, :738)
    return c


def x0_of(c):
    if c["kind"] == "rosenbrock":
        return P.rosenbrock_x0(c["n"])
    r = np.random.default_rng(c["seed"] + 1000)
    return c["x0_scale"] * r.standard_normal(c["n"])


def configure(b, c):
    b = b.with_m(c["m"]).with_max_iterations(c["iters"]).with_epsilon(1e-12).with_max_step_size(c["max_step"])
    b = b.with_initial_step_size(c["h0"]).with_linesearch_gtol(c["gtol"]).with_max_linesearch(c["max_ls"])
    b = b.with_linesearch_algorithm(c["algo"]).with_damping(c["damping"])
    if c["gradient_only"]:
        b = b.with_gradient_only()
    if c["owl"] is not None:
        b = b.with_orthantwise(*c["owl"])
    return b


def oracle_objective(c):
    return {"quadratic": O.quadratic, "logistic": O.logistic, "rosenbrock": O.rosenbrock}[c["kind"]]()


def run_oracle(c, mode=0):
    """-> (rows, x, error code or 0)"""
    O.lib().oracle_set_dot_mode(mode)
    try:
        rows, x = [], x0_of(c)
        try:
            configure(O.lbfgs(), c).minimize(x, oracle_objective(c), lambda p: rows.append(
                (p["niter"], p["neval"], p["ncall"], p["fx"], p["xnorm"], p["gnorm"], p["step"])) and False)
            return rows, x, 0
        except O.OracleError as e:
            return rows, x, e.code
    finally:
        O.lib().oracle_set_dot_mode(0)


def run_product(R, objectives, c):
    dev = {"quadratic": objectives.Quadratic, "logistic": objectives.Logistic, "rosenbrock": objectives.Rosenbrock}[c["kind"]](
        fuse_line_eval=c.get("fuse", 2))
    rows, x = [], x0_of(c)
    try:
        b = configure(R.lbfgs(), c)
        if c.get("vector_free") and c["m"] <= 10:
            b = b.with_vector_free(True)
        b.minimize(x, dev, lambda p: rows.append((p.niter, p.neval, p.ncall, p.fx, p.xnorm, p.gnorm, p.step)) and False)
        return rows, x, 0
    except R.LbfgsError as e:
        return rows, x, e.code
