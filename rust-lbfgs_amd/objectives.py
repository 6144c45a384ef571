"""Device-resident objectives: the `evaluate` closure kept in HBM (lbfgs_hip_objective).

`Quadratic` and `Logistic` are the synthetic workloads of BASELINE.json configs 2-4; their
data is a counter-based hash of the GLOBAL element index (nothing stored, identical on every
rank and on the CPU oracle).  `Rosenbrock` is the reference's default_evaluate (src/lib.rs:79-94).
"""
from . import _ffi
from .api import BuiltinObjective

SEED_QUAD_A, SEED_QUAD_B = 0x5EED0001, 0x5EED0002
SEED_LOGI_A, SEED_LOGI_T = 0x5EED0003, 0x5EED0004


def Quadratic(fuse_line_eval=True):
    """f = sum x_i*(0.5*a_i*x_i - b_i), a_i = 1 + 999*u_i^2 (cond 1e3), b_i = 2*u'_i - 1."""
    return BuiltinObjective(_ffi.OBJ_QUADRATIC, SEED_QUAD_A, SEED_QUAD_B, fuse_line_eval)


def Logistic(fuse_line_eval=True):
    """f = sum log(1+exp(-t_i*a_i*x_i)), a_i = 0.5 + 1.5*u_i, t_i = +-1."""
    return BuiltinObjective(_ffi.OBJ_LOGISTIC, SEED_LOGI_A, SEED_LOGI_T, fuse_line_eval)


def Rosenbrock(fuse_line_eval=True):
    return BuiltinObjective(_ffi.OBJ_ROSENBROCK, 0, 0, fuse_line_eval)
