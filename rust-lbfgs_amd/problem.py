"""`Problem`, `LineSearch`, `Orthantwise`, `signum`: the rest of the reference's public surface.

    prb = Problem(x, default_evaluate(), None)        # core.rs:59   (x is uploaded; vectors live in HBM)
    prb.evaluate()                                     # core.rs:119
    prb.update_search_direction()                      # core.rs:95
    step = 1.0 / prb.search_direction().vec2norm()     # line.rs:28
    ncall, step = LineSearch().find(prb, step)         # line.rs:193  (Rust: find(&mut prb, &mut step) -> Result<usize>)

mirrors the doctest of src/line.rs:8-32.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _ffi
from .api import Context, LbfgsError, Param, _dp, _make_evaluator, _raise


def signum(x: float) -> float:
    """orthantwise.rs:174-180: NaN and +-0 -> 0, else the sign."""
    if math.isnan(x) or x == 0.0:
        return 0.0
    return math.copysign(1.0, x)


@dataclass
class Orthantwise:
    """orthantwise.rs:19-55"""
    c: float = 1.0
    start: int = 0
    end: Optional[int] = None


class LineSearchAlgorithm:
    """line.rs:39-81"""
    MoreThuente = _ffi.LS_MORETHUENTE
    BacktrackingArmijo = _ffi.LS_BT_ARMIJO
    BacktrackingStrongWolfe = _ffi.LS_BT_STRONGWOLFE
    BacktrackingWolfe = _ffi.LS_BT_WOLFE


@dataclass
class LineSearch:
    """line.rs:91-163 (defaults :151-162)"""
    algorithm: int = LineSearchAlgorithm.MoreThuente
    ftol: float = 1e-4
    gtol: float = 0.9
    xtol: float = 2.220446049250313e-16
    min_step: float = 1e-20
    max_step: float = 1e20
    max_linesearch: int = 20
    gradient_only: bool = False

    def find(self, prb: "Problem", step: float):
        """line.rs:193-223.  Returns (ncall, step); a failed search reverts the problem and returns ncall = 0."""
        p = Param()
        p.ls_algorithm, p.gradient_only = self.algorithm, int(self.gradient_only)
        p.ftol, p.gtol, p.xtol = self.ftol, self.gtol, self.xtol
        p.min_step, p.max_step, p.max_linesearch = self.min_step, self.max_step, self.max_linesearch
        prb._check(prb._L.lbfgs_problem_set_linesearch(prb._h, C.byref(p)))
        s, n = C.c_double(step), C.c_uint64()
        prb._check(prb._L.lbfgs_line_search(prb._h, C.byref(s), C.byref(n)))
        return n.value, s.value


class Problem:
    """core.rs:10-218 with the vectors resident on the device."""

    def __init__(self, x, evaluate, owlqn: Optional[Orthantwise] = None, *, ctx=None, device=0):
        L = _ffi.load()
        self._L = L
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._own_ctx = ctx is None
        self.ctx = ctx if ctx is not None else Context(len(x), device=device)
        p = Param()
        L.lbfgs_param_default(C.byref(p))
        if owlqn is not None:
            p.orthantwise, p.owl_c, p.owl_start = 1, owlqn.c, owlqn.start
            p.owl_end = -1 if owlqn.end is None else owlqn.end
        self._ev, self._keep, self._holder = _make_evaluator(evaluate, self.ctx)
        self._h = C.c_void_p()
        rc = L.lbfgs_problem_new(C.byref(self._h), self.ctx._h, C.byref(p), _dp(x), C.byref(self._ev))
        if rc != 0:
            _raise(rc, L.lbfgs_state_error(None).decode())

    def _check(self, rc):
        if rc != 0:
            msg = self._L.lbfgs_state_error(self._h).decode()
            exc = self._holder.get("exc")
            if rc == _ffi.ERR_EVALUATE and exc is not None:
                self._holder["exc"] = None
                raise LbfgsError(rc, f"{msg}: {exc!r}") from exc
            _raise(rc, msg)

    def close(self):
        if self._h:
            self._L.lbfgs_state_free(self._h)
            self._h = C.c_void_p()
        if self._own_ctx:
            self.ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # --- core.rs methods, same names
    def evaluate(self):
        self._check(self._L.lbfgs_problem_evaluate(self._h))

    def update_search_direction(self):
        self._check(self._L.lbfgs_problem_update_search_direction(self._h))

    def search_direction(self):
        """`&[f64]` in the reference; here the device vector d itself (borrowed), with the LbfgsMath methods."""
        from .hotpath import _Borrowed

        return _Borrowed(self.ctx, self._L.lbfgs_state_vec(self._h, _ffi.VEC_D))

    def search_direction_mut(self):
        return self.search_direction()

    def dginit(self) -> float:
        out = C.c_double()
        self._check(self._L.lbfgs_problem_dginit(self._h, C.byref(out)))
        return out.value

    def dg_unchecked(self) -> float:
        out = C.c_double()
        self._check(self._L.lbfgs_problem_dg_unchecked(self._h, C.byref(out)))
        return out.value

    def save_state(self):
        self._check(self._L.lbfgs_problem_save_state(self._h))

    def revert(self):
        self._check(self._L.lbfgs_problem_revert(self._h))

    def take_line_step(self, step: float):
        self._check(self._L.lbfgs_problem_take_line_step(self._h, float(step)))

    def update_orthant_new_point(self):
        self._check(self._L.lbfgs_problem_update_orthant_new_point(self._h))

    def constrain_search_direction(self):
        self._check(self._L.lbfgs_problem_constrain_search_direction(self._h))

    def _norms(self):
        xn, gn = C.c_double(), C.c_double()
        self._check(self._L.lbfgs_problem_norms(self._h, C.byref(xn), C.byref(gn)))
        return xn.value, gn.value

    def xnorm(self) -> float:
        return self._norms()[0]

    def gnorm(self) -> float:
        return self._norms()[1]

    def _status(self):
        fx, ne, ev, ow = C.c_double(), C.c_uint64(), C.c_int(), C.c_int()
        self._check(self._L.lbfgs_problem_status(self._h, C.byref(fx), C.byref(ne), C.byref(ev), C.byref(ow)))
        return fx.value, ne.value, bool(ev.value), bool(ow.value)

    @property
    def fx(self) -> float:
        return self._status()[0]

    def number_of_evaluation(self) -> int:
        return self._status()[1]

    def evaluated(self) -> bool:
        return self._status()[2]

    def orthantwise(self) -> bool:
        return self._status()[3]

    # vectors (host copies)
    def _download(self, which):
        out = np.zeros(self.ctx.n_local)
        self._check(self._L.lbfgs_state_download(self._h, which, _dp(out)))
        return out

    @property
    def x(self):
        return self._download(_ffi.VEC_X)

    @property
    def gx(self):
        return self._download(_ffi.VEC_GX)
