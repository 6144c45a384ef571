"""Thin Python handles on the fused hot-path operators of include/lbfgs_hip.h.

These are the calls a Rust shim would make from `lbfgs_two_loop_recursion`,
`IterationData::update`, `Problem::take_line_step` and the OWL-QN methods;
the parity tests and bench.py drive them directly.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .api import BuiltinObjective, Context, _dp
from .math import DeviceVec


class _Borrowed(DeviceVec):
    """A DeviceVec view of a vector owned by a History (not freed by us)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._L = ctx._L
        self._h = C.c_void_p(handle)

    def free(self):
        pass


class History:
    """The m (s, y, ys, alpha) corrections: `Vec<IterationData>` of lbfgs.rs:437,607-627."""

    def __init__(self, ctx: Context, m: int):
        self.ctx, self.m, self._L = ctx, m, ctx._L
        self._h = C.c_void_p()
        ctx.check(self._L.lbfgs_hip_history_create(ctx._h, m, C.byref(self._h)))

    def free(self):
        if self._h:
            self._L.lbfgs_hip_history_destroy(self._h)
            self._h = C.c_void_p()

    def s(self, slot):
        return _Borrowed(self.ctx, self._L.lbfgs_hip_history_s(self._h, slot))

    def y(self, slot):
        return _Borrowed(self.ctx, self._L.lbfgs_hip_history_y(self._h, slot))

    def scalars(self):
        ys, al = np.zeros(self.m), np.zeros(self.m)
        self.ctx.check(self._L.lbfgs_hip_history_scalars_read(self._h, _dp(ys), _dp(al)))
        return ys, al

    def set_scalars(self, ys=None, alpha=None):
        ysp = _dp(np.ascontiguousarray(ys, dtype=np.float64)) if ys is not None else None
        alp = _dp(np.ascontiguousarray(alpha, dtype=np.float64)) if alpha is not None else None
        self.ctx.check(self._L.lbfgs_hip_history_scalars_write(self._h, ysp, alp))

    def update(self, slot, x, xp, g, gp, step=1.0, damping=False, out_slot=6):
        """IterationData::update, vector part (lbfgs.rs:640-673)."""
        self.ctx.check(self._L.lbfgs_hip_history_update(self._h, slot, x._h, xp._h, g._h, gp._h, float(step),
                                                        int(damping), out_slot))

    def update_from_step(self, slot, obj, x, xp, d, t, g, gp, step=1.0, damping=False, out_slot=6):
        """x = xp + t*d ; g = grad f(x) ; then `update` in the same pass (element-wise objectives).  3r 4w."""
        o = obj.c_struct(self.ctx)
        self.ctx.check(self._L.lbfgs_hip_history_update_from_step(self._h, slot, C.byref(o), x._h, xp._h, d._h, float(t),
                                                                  g._h, gp._h, float(step), int(damping), out_slot))

    def damp(self, slot, gp, step, theta):
        self.ctx.check(self._L.lbfgs_hip_history_damp(self._h, slot, gp._h, float(step), float(theta)))

    def two_loop(self, d, g, k, end, gamma_num_slot=7, gamma_den_slot=8, dnorm_slot=12):
        """d = -g; lbfgs_two_loop_recursion (lbfgs.rs:569-604); board[dnorm_slot] = ||d||^2.  Returns new end."""
        ne = C.c_int()
        self.ctx.check(self._L.lbfgs_hip_two_loop(self._h, d._h, g._h, k, end, gamma_num_slot, gamma_den_slot,
                                                  dnorm_slot, C.byref(ne)))
        return ne.value

    def two_loop_from(self, d, g, k, end, first_dot_slot, gamma_num_slot=7, gamma_den_slot=8, dnorm_slot=13):
        """two_loop starting from the alpha_0 numerator that `update` left at out_slot+6."""
        ne = C.c_int()
        self.ctx.check(self._L.lbfgs_hip_two_loop_from(self._h, d._h, g._h, k, end, gamma_num_slot, gamma_den_slot,
                                                       dnorm_slot, first_dot_slot, C.byref(ne)))
        return ne.value

    def two_loop_owlqn(self, d, pg, k, end, start, end_, gamma_num_slot=7, gamma_den_slot=8, dnorm_slot=13):
        """two_loop under OWL-QN with the orthant projection of d folded into the last step."""
        ne = C.c_int()
        self.ctx.check(self._L.lbfgs_hip_two_loop_owlqn(self._h, d._h, pg._h, k, end, gamma_num_slot, gamma_den_slot,
                                                        dnorm_slot, start, end_, C.byref(ne)))
        return ne.value

    def two_loop_gram(self, d, g, k, end, gamma_num_slot=7, gamma_den_slot=8, dnorm_slot=12):
        """Vector-free (Gram) variant [extension]: call after every history update."""
        ne = C.c_int()
        self.ctx.check(self._L.lbfgs_hip_two_loop_gram(self._h, d._h, g._h, k, end, gamma_num_slot, gamma_den_slot,
                                                       dnorm_slot, C.byref(ne)))
        return ne.value

    def two_loop_unfused(self, d, k, end, gamma_num_slot=7, gamma_den_slot=8):
        ne = C.c_int()
        self.ctx.check(self._L.lbfgs_hip_two_loop_unfused(self._h, d._h, k, end, gamma_num_slot, gamma_den_slot,
                                                          C.byref(ne)))
        return ne.value


def line_step(x, xp, d, step, wp=None, start=0, end=0):
    """Problem::take_line_step (core.rs:155-164)."""
    x.ctx.check(x._L.lbfgs_hip_line_step(x._h, xp._h, d._h, float(step), wp._h if wp is not None else None,
                                         start, end))


def norms_sq(x, g, out_slot=14):
    x.ctx.check(x._L.lbfgs_hip_norms_sq(x._h, g._h, out_slot))


def owlqn_post_eval(x, g, pg, c, start, end, out_slot=2):
    """x1norm + compute_pseudo_gradient + norms (core.rs:123-126)."""
    x.ctx.check(x._L.lbfgs_hip_owlqn_post_eval(x._h, g._h, pg._h, float(c), start, end, out_slot))


def orthant_select(wp, xp, pg):
    """Problem::update_orthant_new_point (core.rs:167-180)."""
    wp.ctx.check(wp._L.lbfgs_hip_orthant_select(wp._h, xp._h, pg._h))


def constrain_direction(d, pg, start, end, out_slot=13):
    """Orthantwise::constrain_search_direction (orthantwise.rs:140-161)."""
    d.ctx.check(d._L.lbfgs_hip_constrain_direction(d._h, pg._h, start, end, out_slot))


def objective_eval(obj: BuiltinObjective, x, g, out_slot=0):
    o = obj.c_struct(x.ctx)
    x.ctx.check(x._L.lbfgs_hip_objective_eval(C.byref(o), x._h, g._h, out_slot))


def objective_line_eval(obj: BuiltinObjective, x, xp, d, step, g, out_slot=0):
    o = obj.c_struct(x.ctx)
    x.ctx.check(x._L.lbfgs_hip_objective_line_eval(C.byref(o), x._h, xp._h, d._h, float(step), g._h, out_slot))


def objective_line_probe(obj: BuiltinObjective, xp, d, step, out_slot=0):
    """f(xp + step*d) and grad.d with nothing written (element-wise objectives).  2r 0w."""
    o = obj.c_struct(xp.ctx)
    xp.ctx.check(xp._L.lbfgs_hip_objective_line_probe(C.byref(o), xp._h, d._h, float(step), out_slot))


def objective_owlqn_line_eval(obj: BuiltinObjective, x, xp, d, step, wp, g, pg, c, start, end, out_slot=0):
    """One OWL-QN trial in one pass: projected line step + evaluate + x1norm + pseudo-gradient (+ g.d)."""
    o = obj.c_struct(x.ctx)
    x.ctx.check(x._L.lbfgs_hip_objective_owlqn_line_eval(C.byref(o), x._h, xp._h, d._h, float(step), wp._h, g._h, pg._h,
                                                         float(c), start, end, out_slot))


def objective_owlqn_first_trial(obj: BuiltinObjective, x, xp, d, step, wp, g, pg, c, start, end, out_slot=0):
    """The first trial of an OWL-QN search with update_orthant_new_point (core.rs:167-180) folded in: wp is an OUTPUT, formed
    from xp and the pseudo-gradient pg holds on entry; pg then receives the trial point's.  3r 4w."""
    o = obj.c_struct(x.ctx)
    x.ctx.check(x._L.lbfgs_hip_objective_owlqn_first_trial(C.byref(o), x._h, xp._h, d._h, float(step), wp._h, g._h, pg._h,
                                                           float(c), start, end, out_slot))


def objective_owlqn_trial_update(obj: BuiltinObjective, hist: History, slot, x, xp, d, step, wp, first, g, gp, pg, c, start, end,
                                 out_slot=0, upd_slot=6):
    """An OWL-QN trial that also does IterationData::update for its point (lbfgs.rs:640-656): s, y into history slot `slot`,
    ||s||^2, y.s, y.y at board[upd_slot ..].  `first`: the trial also forms the orthant (objective_owlqn_first_trial).  4r 6w."""
    o = obj.c_struct(x.ctx)
    x.ctx.check(x._L.lbfgs_hip_objective_owlqn_trial_update(C.byref(o), hist._h, slot, x._h, xp._h, d._h, float(step), wp._h,
                                                            int(bool(first)), g._h, gp._h, pg._h, float(c), start, end,
                                                            out_slot, upd_slot))
