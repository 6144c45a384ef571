"""CPU oracle vs the reference's own known answers ("what pins results").

These are the tests that PIN the oracle: every number compared against here is
held by the reference's own tests (tests/golden/reference_known_answers.json).
"""
import json
import os

import numpy as np

from oracle import oracle as O
from tests import problems as P

KA = json.load(open(os.path.join(P.GOLDEN, "reference_known_answers.json")))


def test_math_primitives_exact():
    """src/math.rs:84-122 test_lbfgs_math, assert_eq! exactness."""
    ka = KA["math_rs_84_122"]
    x = np.array(ka["vecadd"]["x"])
    y = np.array(ka["vecadd"]["y"])
    O.vecadd(y, x, ka["vecadd"]["c"])
    assert y.tolist() == ka["vecadd"]["expect"]
    assert O.vecdot(y, x) == ka["vecdot"]["expect"]
    O.vecscale(y, ka["vecscale"]["c"])
    assert y.tolist() == ka["vecscale"]["expect"]
    z = y.copy()
    O.vecdiff(z, x, y)
    assert z.tolist() == ka["vecdiff_x_minus_y"]["expect"]
    O.veccpy(y, x)
    assert y.tolist() == x.tolist()
    O.vecncpy(y, x)
    assert y.tolist() == ka["vecncpy"]["expect"]
    assert O.vec2norm(np.array([3.0, 4.0])) == 5.0
    assert O.vec2norminv(np.array([3.0, 4.0])) == 0.2


def test_rosenbrock_defaults_asserts():
    """tests/simple.rs:17-40 with the crate's current defaults (max_step_size = 1)."""
    ka = KA["simple_rs_rosenbrock"]["assert_37_40"]
    x = P.rosenbrock_x0()
    rep = O.lbfgs().minimize(x, O.rosenbrock())
    assert abs(rep["fx"] - ka["fx"]) <= ka["abs_tol"]
    assert np.all(np.abs(x - ka["x_all"]) <= ka["abs_tol"])
    # python closure and C objective agree bit for bit
    x2 = P.rosenbrock_x0()
    rep2 = O.lbfgs().minimize(x2, P.rosenbrock)
    assert rep2 == rep and np.array_equal(x, x2)


def test_rosenbrock_and_owlqn_17_digits():
    """tests/simple.rs:33-35 and :48-50 -- bit-exact known answers (step clamp neutralised)."""
    ka = KA["simple_rs_rosenbrock"]["comment_33_35"]
    x = P.rosenbrock_x0()
    last = {}
    rep = O.lbfgs().with_max_step_size(1e20).minimize(x, O.rosenbrock(), lambda p: last.update(p) and False)
    assert last["niter"] == 38  # reference's "Iteration 37" + the no-op first propagate
    assert rep["fx"] == ka["fx"]
    assert x[0] == ka["x0"] and x[1] == ka["x1"]
    assert rep["xnorm"] == ka["xnorm"] and rep["gnorm"] == ka["gnorm"]
    assert last["step"] == ka["step"]

    kb = KA["simple_rs_owlqn"]
    ow = kb["orthantwise"]
    last = {}
    rep = (O.lbfgs().with_max_step_size(1e20).with_orthantwise(ow["c"], ow["start"], ow["end"])
           .minimize(x, O.rosenbrock(), lambda p: last.update(p) and False))
    kc = kb["comment_48_50"]
    assert last["niter"] == 172
    assert rep["fx"] == kc["fx"]
    assert x[0] == kc["x0"] and x[1] == kc["x1"]
    assert rep["xnorm"] == kc["xnorm"] and rep["gnorm"] == kc["gnorm"]
    assert last["step"] == kc["step"]


def test_owlqn_defaults_asserts():
    """tests/simple.rs:42-54 with current defaults: OWL-QN continued from the converged x."""
    x = P.rosenbrock_x0()
    O.lbfgs().minimize(x, O.rosenbrock())
    kb = KA["simple_rs_owlqn"]
    ow, ka = kb["orthantwise"], kb["assert_52_54"]
    rep = O.lbfgs().with_orthantwise(ow["c"], ow["start"], ow["end"]).minimize(x, O.rosenbrock())
    assert abs(rep["fx"] - ka["fx"]) <= ka["abs_tol"]
    assert abs(x[0] - ka["x0"]) <= ka["abs_tol"]
    assert abs(x[1] - ka["x1"]) <= ka["abs_tol"]


def test_booth():
    """tests/simple.rs:58-83"""
    ka = KA["simple_rs_booth_81_82"]
    x = np.array(ka["x0"])
    O.lbfgs().minimize(x, P.booth)
    assert np.all(np.abs(x - np.array(ka["expect"])) <= ka["abs_tol"])


def test_poisson_owlqn():
    """tests/owlqn.rs:6-63"""
    ka = KA["owlqn_rs_60"]
    evaluate, ncol = P.poisson_problem()
    x = np.zeros(ncol)
    ow = ka["orthantwise"]
    rep = (O.lbfgs().with_orthantwise(ow["c"], ow["start"], ow["end"]).with_epsilon(ka["epsilon"])
           .minimize(x, evaluate))
    assert abs(rep["fx"] - ka["fx"]) <= ka["abs_tol"]


def test_doc_example_max_iterations():
    """src/lib.rs:38-50: with_max_iterations(5) stops at niter 5 (lbfgs.rs:729)."""
    x = P.rosenbrock_x0()
    seen = []
    O.lbfgs().with_max_iterations(5).minimize(x, O.rosenbrock(), lambda p: seen.append(p["niter"]) and False)
    assert seen == [1, 2, 3, 4, 5]


def test_diagnostic_summation_orders():
    """`oracle_set_dot_mode` 1-4 only re-order sums (the tests' summation-order sensitivity estimate rests on that):
    each mode against the same order written out in Python, mode 0 against the reference's running sum (math.rs:41)."""
    r = np.random.default_rng(5)
    for n in (1, 2, 5, 63, 64, 65, 257, 1000):
        x, y = r.standard_normal(n) * 10.0 ** r.integers(-3, 4, n), r.standard_normal(n)
        t = x * y

        def strided(K):
            acc = [0.0] * K
            for i in range(n):
                acc[i % K] += t[i]
            w = 1
            while w < K:
                for k in range(0, K - w, 2 * w):
                    acc[k] += acc[k + w]
                w *= 2
            return acc[0]

        def pairwise(a):
            if len(a) <= 32:
                s = 0.0
                for v in a:
                    s += v
                return s
            h = len(a) // 2
            return pairwise(a[:h]) + pairwise(a[h:])

        def seq(a):
            s = 0.0
            for v in a:
                s += v
            return s

        want = {0: seq(t), 1: pairwise(list(t)), 2: seq(t[::-1]), 3: strided(4), 4: strided(64)}
        try:
            for mode, w in want.items():
                O.lib().oracle_set_dot_mode(mode)
                assert O.vecdot(x, y) == w, (n, mode)
        finally:
            O.lib().oracle_set_dot_mode(0)
