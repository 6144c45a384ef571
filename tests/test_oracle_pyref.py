"""The C oracle against a SECOND, independent restatement of the reference (oracle/pyref.py, pure Python, written from the Rust
text alone), bit for bit, on the paths no test or vector of the reference holds: Powell damping,
`gradient_only`, the backtracking variants without OWL-QN, m != 6 -- and, for good measure, the pinned ones (More-Thuente,
OWL-QN).  Every iteration's niter / neval / ncall / fx / xnorm / gnorm / step and the final x must be IDENTICAL: the two
transcriptions share no code (C vs Python, arrays vs lists), only the source they restate and IEEE arithmetic.
CPU only; small problems (pure-Python loops)."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pyref as PY


def rosenbrock(x, g):                      # src/lib.rs:79-94
    fx = 0.0
    for i in range(0, len(x), 2):
        t1 = 1.0 - x[i]
        t2 = 10.0 * (x[i + 1] - x[i] * x[i])
        g[i + 1] = 20.0 * t2
        g[i] = -2.0 * (x[i] * g[i + 1] + t1)
        fx += t1 * t1 + t2 * t2
    return float(fx)


def chained_exp(x, g):                     # smooth, non-quadratic, coupled: More-Thuente brackets and interpolates on it
    n = len(x)
    fx = 0.0
    for i in range(n):
        g[i] = 0.0
    for i in range(n):
        e = math.exp(0.3 * float(x[i]))
        fx += e - 0.9 * float(x[i]) * (1.0 + 0.1 * (i % 3))
        g[i] += 0.3 * e - 0.9 * (1.0 + 0.1 * (i % 3))
        if i + 1 < n:
            d = float(x[i + 1]) - float(x[i]) * float(x[i])
            fx += 0.05 * d * d
            g[i + 1] += 0.1 * d
            g[i] += -0.2 * d * float(x[i])
    return float(fx)


def quartic_bowl(x, g):                    # flat bottom: long searches, tiny curvature pairs (exercises Powell damping)
    fx = 0.0
    for i in range(len(x)):
        t = float(x[i]) - 0.5 * (1 + i % 2)
        fx += t * t * t * t + 1e-3 * t * t
        g[i] = 4.0 * t * t * t + 2e-3 * t
    return float(fx)


OBJECTIVES = {"rosenbrock": (rosenbrock, lambda n: [-1.2, 1.0] * (n // 2)),
              "chained_exp": (chained_exp, lambda n: [0.5 - 0.3 * (i % 4) for i in range(n)]),
              "quartic_bowl": (quartic_bowl, lambda n: [2.0 - 0.7 * (i % 5) for i in range(n)])}

CASES = {
    "defaults": {},
    "no_step_clamp": dict(max_step_size=1e20),
    "damping": dict(damping=True),
    "damping_no_clamp": dict(damping=True, max_step_size=1e20),
    "armijo": dict(algo="BacktrackingArmijo"),
    "wolfe": dict(algo="BacktrackingWolfe"),
    "strong_wolfe": dict(algo="BacktrackingStrongWolfe"),
    "strong_wolfe_damped": dict(algo="BacktrackingStrongWolfe", damping=True),
    "gradient_only": dict(gradient_only=True),
    "m3": dict(m=3),
    "m10_damped_wolfe": dict(m=10, damping=True, algo="BacktrackingWolfe"),
    "owlqn": dict(owl=(0.7, 2, None)),
    "owlqn_range_m4": dict(owl=(0.3, 1, 9), m=4),
    "short_searches": dict(max_linesearch=3),
    "short_searches_wolfe": dict(max_linesearch=4, algo="BacktrackingWolfe", max_step_size=1e20),
    "min_step_trips": dict(algo="BacktrackingArmijo", min_step=0.4, max_step_size=1e20),     # validate_step fails -> revert -> "x not changed"
    "loose_curvature": dict(gtol=0.1, ftol=1e-3),
    "h0": dict(initial_inverse_hessian=0.25, max_step_size=3.0),
}


def configure(b, c, python):
    """the same settings on the C oracle's builder (python = False) and on pyref's parameter object"""
    if python:
        ls = b.linesearch
        b.m = c.get("m", 6)
        b.max_iterations, b.epsilon = 40, c.get("epsilon", 1e-7)
        b.damping = c.get("damping", False)
        b.max_step_size = c.get("max_step_size", 1.0)
        b.initial_inverse_hessian = c.get("initial_inverse_hessian", 1.0)
        ls.algorithm = c.get("algo", "MoreThuente")
        ls.max_linesearch = c.get("max_linesearch", 20)
        ls.min_step = c.get("min_step", 1e-20)
        ls.ftol, ls.gtol = c.get("ftol", 1e-4), c.get("gtol", 0.9)
        if c.get("gradient_only"):
            b.with_gradient_only()
        if c.get("owl"):
            b.orthantwise = PY.Orthantwise(*c["owl"])
        return b
    b = b.with_m(c.get("m", 6)).with_max_iterations(40).with_epsilon(c.get("epsilon", 1e-7)).with_damping(c.get("damping", False))
    b = b.with_max_step_size(c.get("max_step_size", 1.0)).with_initial_step_size(c.get("initial_inverse_hessian", 1.0))
    b = b.with_linesearch_algorithm(c.get("algo", "MoreThuente")).with_max_linesearch(c.get("max_linesearch", 20))
    b = b.with_linesearch_min_step(c.get("min_step", 1e-20)).with_linesearch_ftol(c.get("ftol", 1e-4)).with_linesearch_gtol(c.get("gtol", 0.9))
    if c.get("gradient_only"):
        b = b.with_gradient_only()
    if c.get("owl"):
        b = b.with_orthantwise(*c["owl"])
    return b


KEYS = ("niter", "neval", "ncall", "fx", "xnorm", "gnorm", "step")


@pytest.mark.parametrize("objective", sorted(OBJECTIVES))
@pytest.mark.parametrize("case", sorted(CASES))
def test_c_oracle_and_python_restatement_agree_bit_for_bit(case, objective):
    fn, x0 = OBJECTIVES[objective]
    n = 12
    c = CASES[case]
    # --- the C oracle
    xo = np.array(x0(n), dtype=np.float64)
    rows_o, err_o = [], None
    try:
        configure(O.lbfgs(), c, False).minimize(xo, fn, lambda p: rows_o.append(tuple(p[k] for k in KEYS)) and False)
    except O.OracleError as e:
        err_o = str(e)
    # --- the Python restatement
    xp = [float(v) for v in x0(n)]
    rows_p, err_p = configure(PY.Lbfgs(), c, True).minimize(xp, fn)
    rows_p = [tuple(p[k] for k in KEYS) for p in rows_p]
    assert len(rows_o) == len(rows_p) and len(rows_o) >= 1, (len(rows_o), len(rows_p), err_o, err_p)
    for a, b in zip(rows_o, rows_p):
        assert a == b, (case, objective, a, b)            # EXACT: integers and every float bit for bit
    assert (err_o is None) == (err_p is None), (err_o, err_p)
    assert xo.tolist() == xp
    if err_o is not None:                                # (which error: the same one)
        assert ("x not changed" in err_o) == ("x not changed" in err_p) and ("gx not changed" in err_o) == ("gx not changed" in err_p)


def test_the_cases_reach_the_paths_they_are_named_for():
    """The comparison above would be hollow if the runs never took the unpinned branches: count them in the restatement."""
    hits = {"case1": 0, "case2": 0, "inc": 0, "dec": 0, "mt_modified": 0, "mt_cases": set()}
    orig_update, orig_uti = PY.IterationData.update, PY.update_trial_interval

    def update(self, x, xp, gx, gp, step, damping):
        if damping:
            s = [a - b for a, b in zip(x, xp)]
            y = [a - b for a, b in zip(gx, gp)]
            ys = PY.vecdot(y, s)
            sbs = PY.vecdot(s, [-step * v for v in gp])
            hits["case1"] += ys < 0.4 * sbs
            hits["case2"] += ys > 4.0 * sbs
        return orig_update(self, x, xp, gx, gp, step, damping)

    def uti(S, ft, dt, tmin, tmax):
        fx, dx = S["fx"], S["dx"]
        hits["mt_cases"].add(1 if fx < ft else 2 if dt * (dx / abs(dx)) < 0.0 else 3 if abs(dt) < abs(dx) else 4)
        return orig_uti(S, ft, dt, tmin, tmax)

    PY.IterationData.update, PY.update_trial_interval = update, uti
    try:
        for objective, (fn, x0) in OBJECTIVES.items():
            for case, c in CASES.items():
                rows, _ = configure(PY.Lbfgs(), c, True).minimize([float(v) for v in x0(12)], fn)
                if c.get("algo", "").startswith("Backtracking") or c.get("gradient_only"):
                    hits["dec"] += sum(1 for r in rows if r["ncall"] > 1)
    finally:
        PY.IterationData.update, PY.update_trial_interval = orig_update, orig_uti
    assert hits["case1"] > 0, hits                      # Powell damping case 1 (y replaced) fires
    assert hits["dec"] > 0, hits                        # backtracking searches that needed more than one trial
    assert {1, 2, 3} <= hits["mt_cases"], hits          # MCSTEP's cases: higher value / sign change / smaller derivative
