"""Host-side objectives used by the parity tests (the reference's own test problems).

Each is an `evaluate(x, gx) -> fx` closure in the reference's shape
(E: FnMut(&[f64], &mut [f64]) -> Result<f64>, src/core.rs:12).
"""
import csv
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def lj38_x0():
    """examples/lj.rs:72-110: the LJ38 start coordinates of the reference's example (tests/golden/lj38_positions.json)."""
    import json

    d = json.load(open(os.path.join(GOLDEN, "lj38_positions.json")))
    x = np.array(d["positions"], dtype=np.float64)
    assert x.shape == (3 * d["natoms"],) == (114,)
    return x


def rosenbrock_x0(n=100):
    """tests/simple.rs:24-28"""
    x = np.zeros(n)
    x[0::2] = -1.2
    x[1::2] = 1.0
    return x


def rosenbrock(x, gx):
    """src/lib.rs:79-94 default_evaluate (sequential accumulation of fx)."""
    fx = 0.0
    for i in range(0, len(x), 2):
        t1 = 1.0 - x[i]
        t2 = 10.0 * (x[i + 1] - x[i] * x[i])
        gx[i + 1] = 20.0 * t2
        gx[i] = -2.0 * (x[i] * gx[i + 1] + t1)
        fx += t1 * t1 + t2 * t2
    return fx


def booth(x, gx):
    """tests/simple.rs:67-75"""
    x1, x2 = x[0], x[1]
    fx = (x1 + 2.0 * x2 - 7.0) ** 2 + (2.0 * x1 + x2 - 5.0) ** 2
    gx[0] = 10.0 * x1 + 8.0 * x2 - 34.0
    gx[1] = 8.0 * x1 + 10.0 * x2 - 38.0
    return fx


def _read_csv(path):
    """tests/owlqn.rs:66-83: skip the header row and the index column."""
    out = []
    with open(path) as f:
        rd = csv.reader(f)
        next(rd)
        for row in rd:
            out.append([float(v) for v in row[1:]])
    return np.array(out)


def poisson_problem():
    """tests/owlqn.rs:9-44: Poisson regression NLL on the 500x21 fixture (prec = 0)."""
    X = _read_csv(os.path.join(GOLDEN, "poisson_x.csv"))
    y = _read_csv(os.path.join(GOLDEN, "poisson_y.csv"))[:, 0]
    assert X.shape == (500, 21) and y.shape == (500,)

    def evaluate(x, gx):
        xbeta = X @ x
        e = np.exp(xbeta)
        fx = -float(np.sum(y * xbeta - e))
        gx[:] = -(X.T @ (y - e))
        return fx

    return evaluate, 21
