"""Generate tests/golden/oracle_traces.json from the CPU oracle (run AFTER tests/test_oracle_golden.py passes).

The traces are the committed expected values for the GPU trajectory tests: per iteration f, ||x||,
||g||, step, ncall, neval and the first/last 4 components of the search direction d.

Tolerance of a FREE-RUNNING trajectory: the GPU sums in a fixed tree, the reference sequentially
(src/math.rs:41).  Both are valid f64 evaluations of the same dot products, but an L-BFGS run
amplifies a 1e-16 perturbation as it goes.  Each case therefore carries `rtol`: 1e-10, or -- when
the oracle itself moves by more than that under a mere change of summation order (`oracle_set_dot_mode`:
every sum pairwise, and every sum from its last term down; the LARGER of the two deviations counts -- one
perturbed run is a sample of size one) -- 20x that measured sensitivity.  The strict 1e-10 bar on
identical inputs is enforced separately by the step-locked test (tests/test_gpu_step_locked.py).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import problems as P  # noqa: E402

HEAD = 4


def configure(b, spec):
    for name, args in spec:
        b = getattr(b, name)(*args)
    return b


def oracle_eval(kind):
    return {
        "rosenbrock_closure": O.rosenbrock, "rosenbrock": O.rosenbrock, "quadratic": O.quadratic,
        "quadratic_unfused": O.quadratic, "logistic": O.logistic, "booth": lambda: P.booth,
        "poisson": lambda: P.poisson_problem()[0],
    }[kind]()


def run(spec, evaluate, x0, max_rows):
    x = x0.copy()
    st = configure(O.lbfgs(), spec).build(x, oracle_eval(evaluate))
    rows = []
    err = None
    while len(rows) < max_rows and not st.is_converged():
        try:
            p = st.propagate()
        except O.OracleError as e:
            err = e.code
            break
        d = st.vec("d")
        rows.append(dict(niter=int(p["niter"]), neval=int(p["neval"]), ncall=int(p["ncall"]), fx=p["fx"],
                         xnorm=p["xnorm"], gnorm=p["gnorm"], step=p["step"],
                         d_head=d[:HEAD].tolist(), d_tail=d[max(HEAD, len(d) - HEAD):].tolist(),
                         dnorm_inf=float(np.max(np.abs(d)))))
    conv = bool(st.is_converged()) if err is None else False
    st.close()
    return rows, conv, err


def sensitivity(rows_a, rows_b, f_floor, g_floor):
    """largest scaled difference between two traces of the same case"""
    worst = 0.0
    for a, b in zip(rows_a, rows_b):
        if (a["ncall"], a["neval"]) != (b["ncall"], b["neval"]):
            return float("inf")
        worst = max(worst, abs(a["fx"] - b["fx"]) / max(abs(a["fx"]), f_floor),
                    abs(a["gnorm"] - b["gnorm"]) / max(a["gnorm"], g_floor),
                    abs(a["xnorm"] - b["xnorm"]) / max(a["xnorm"], 1e-300),
                    abs(a["step"] - b["step"]) / abs(a["step"]),
                    float(np.max(np.abs(np.array(a["d_head"] + a["d_tail"]) - np.array(b["d_head"] + b["d_tail"]))))
                    / max(a["dnorm_inf"], g_floor))
    return worst


def main():
    conv_x = P.rosenbrock_x0()
    O.lbfgs().with_max_step_size(1e20).minimize(conv_x, O.rosenbrock())
    cases = {
        "rosen_closure_default": dict(n=100, x0_kind="rosenbrock", evaluate="rosenbrock_closure", builder=[], max_rows=25),
        "rosen_closure_unclamped": dict(n=100, x0_kind="rosenbrock", evaluate="rosenbrock_closure",
                                        builder=[["with_max_step_size", [1e20]]], max_rows=28),
        "rosen_owlqn_from_converged": dict(n=100, x0=conv_x.tolist(), evaluate="rosenbrock_closure",
                                           builder=[["with_max_step_size", [1e20]], ["with_orthantwise", [1.0, 0, 99]]],
                                           max_rows=60),
        "rosen_builtin_m10": dict(n=1000, x0_kind="rosenbrock", evaluate="rosenbrock",
                                  builder=[["with_m", [10]]], max_rows=25),
        "quadratic_m7_morethuente": dict(n=4096, x0_kind="zeros", evaluate="quadratic",
                                         builder=[["with_m", [7]], ["with_epsilon", [0.0]], ["with_max_iterations", [40]]],
                                         max_rows=40),
        "quadratic_m10_armijo_oddn": dict(n=4099, x0_kind="zeros", evaluate="quadratic_unfused",
                                          builder=[["with_m", [10]], ["with_linesearch_algorithm", ["BacktrackingArmijo"]],
                                                   ["with_max_iterations", [40]]], max_rows=40),
        "quadratic_damped_strongwolfe": dict(n=4096, x0_kind="zeros", evaluate="quadratic",
                                             builder=[["with_damping", [True]],
                                                      ["with_linesearch_algorithm", ["BacktrackingStrongWolfe"]],
                                                      ["with_max_iterations", [30]]], max_rows=30),
        "logistic_owlqn_m6": dict(n=4096, x0_kind="zeros", evaluate="logistic",
                                  builder=[["with_orthantwise", [0.5, 0, None]], ["with_max_iterations", [40]]],
                                  max_rows=40),
        "booth": dict(n=2, x0=[-1.2, 1.0], evaluate="booth", builder=[], max_rows=8),
        "poisson_owlqn": dict(n=21, x0_kind="zeros", evaluate="poisson",
                              builder=[["with_orthantwise", [1.0, 1, 21]], ["with_epsilon", [1e-4]]], max_rows=40),
    }
    out = {}
    for name, c in cases.items():
        n = c["n"]
        x0 = np.array(c["x0"]) if "x0" in c else (P.rosenbrock_x0(n) if c["x0_kind"] == "rosenbrock" else np.zeros(n))
        O.lib().oracle_set_dot_mode(0)
        rows, conv, err = run(c["builder"], c["evaluate"], x0, c["max_rows"])
        assert err is None, (name, err)
        perturbed = []
        for mode in (1, 2):  # every sum pairwise / sequential from the last term down
            O.lib().oracle_set_dot_mode(mode)
            perturbed.append(run(c["builder"], c["evaluate"], x0, c["max_rows"])[0])
        O.lib().oracle_set_dot_mode(0)
        f_floor = 1e-6 * max(abs(rows[0]["fx"]), 1e-300)
        g_floor = 1e-6 * rows[0]["gnorm"]
        # keep the prefix of the run over which a change of summation order does not flip a discrete
        # line-search decision and stays below 1e-9 (the rest is chaos, not arithmetic)
        keep = 0
        sens = 0.0
        nfull = len(rows)
        for i in range(1, len(rows) + 1):
            s = max(sensitivity(rows[:i], rp[:i], f_floor, g_floor) if len(rp) >= i else float("inf") for rp in perturbed)
            if s > 5e-10:
                break
            keep, sens = i, s
        rows = rows[:keep]
        rtol = max(1e-10, 20.0 * sens)
        entry = {k: v for k, v in c.items() if k != "max_rows"}
        entry.update(rows=rows, rtol=rtol, f_floor=f_floor, g_floor=g_floor, order_sensitivity=sens,
                     converged_after=(bool(conv) if keep == nfull else None))
        out[name] = entry
        print(f"{name:32s} rows={len(rows):3d} order_sensitivity={sens:.2e} rtol={rtol:.1e} conv={entry['converged_after']}")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_traces.json"), "w") as f:
        json.dump(dict(_generator="tests/golden/make_golden.py", cases=out), f)


if __name__ == "__main__":
    main()
