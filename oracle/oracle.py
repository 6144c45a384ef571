"""ctypes binding of the CPU oracle (oracle/liblbfgs_oracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package ``rust-lbfgs_amd/`` never does.

The API mirrors the reference crate's builder (src/lbfgs.rs:185-384) closely
enough that oracle-side tests read like tests/simple.rs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LBFGS_SANITIZE=1 (tests/test_sanitizers_cpu.py): the -fsanitize=address,undefined build of the same sources
_LIB_NAME = "liblbfgs_oracle_asan.so" if os.environ.get("LBFGS_SANITIZE") == "1" else "liblbfgs_oracle.so"
_LIB_PATH = os.path.join(_HERE, _LIB_NAME)

LS_MORETHUENTE, LS_BT_ARMIJO, LS_BT_STRONGWOLFE, LS_BT_WOLFE = 0, 1, 2, 3

ERRORS = {
    0: "ok",
    -1: "evaluate failed",
    -2: "A logic error (negative line-search step) occurred.",
    -3: "Gradient only optimization is incompatible with MoreThuente line search.",
    -4: "x not changed",
    -5: "gx not changed",
    -6: "invalid norm value",
    -20: "invalid start for orthantwise",
    -21: "invalid direction vector after constraints",
    -30: "allocation failure",
}


class OracleError(RuntimeError):
    def __init__(self, code: int):
        super().__init__(f"oracle status {code}: {ERRORS.get(code, '?')}")
        self.code = code


def source_id() -> str:
    """content hash of the oracle's sources and Makefile (16 hex digits)"""
    import hashlib

    h = hashlib.sha256()
    for f in ("lbfgs_oracle.c", "objectives.c", "lbfgs_oracle.h", "Makefile"):
        h.update(f.encode() + b"\0")
        with open(os.path.join(_HERE, f), "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def embedded_id(lib: str):
    import re

    try:
        with open(lib, "rb") as fh:
            m = re.search(rb"LBFGS_ORACLE_BUILD_ID=([0-9a-f]{16})", fh.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (gcc -O2 -ffp-contract=off).  Rebuilt when the library does not carry the
    hash of the checked-out sources (never by modification time: prebuilt libraries travel with snapshots)."""
    want = source_id()
    if force or embedded_id(_LIB_PATH) != want:
        subprocess.run(["make", "-C", _HERE, "-B", _LIB_NAME, "BUILD_ID=" + want], check=True, capture_output=True)
    return _LIB_PATH


class Param(C.Structure):
    _fields_ = [
        ("m", C.c_uint64),
        ("epsilon", C.c_double),
        ("past", C.c_uint64),
        ("delta", C.c_double),
        ("max_iterations", C.c_uint64),
        ("max_evaluations", C.c_uint64),
        ("initial_inverse_hessian", C.c_double),
        ("max_step_size", C.c_double),
        ("damping", C.c_int32),
        ("constrain_step_size", C.c_int32),
        ("ls_algorithm", C.c_int32),
        ("gradient_only", C.c_int32),
        ("ftol", C.c_double),
        ("gtol", C.c_double),
        ("xtol", C.c_double),
        ("min_step", C.c_double),
        ("max_step", C.c_double),
        ("max_linesearch", C.c_uint64),
        ("orthantwise", C.c_int32),
        ("_pad", C.c_int32),
        ("owl_c", C.c_double),
        ("owl_start", C.c_uint64),
        ("owl_end", C.c_int64),
    ]


class Progress(C.Structure):
    _fields_ = [
        ("x", C.POINTER(C.c_double)),
        ("gx", C.POINTER(C.c_double)),
        ("fx", C.c_double),
        ("xnorm", C.c_double),
        ("gnorm", C.c_double),
        ("step", C.c_double),
        ("niter", C.c_uint64),
        ("neval", C.c_uint64),
        ("ncall", C.c_uint64),
    ]


class Report(C.Structure):
    _fields_ = [("fx", C.c_double), ("xnorm", C.c_double), ("gnorm", C.c_double), ("neval", C.c_uint64)]


class HashedObj(C.Structure):
    _fields_ = [("global_offset", C.c_uint64), ("seed_a", C.c_uint64), ("seed_b", C.c_uint64)]


EVAL_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t,
                      C.POINTER(C.c_int))
PROGRESS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Progress))

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.oracle_param_default.argtypes = [C.POINTER(Param)]
    L.oracle_set_dot_mode.argtypes = [C.c_int]
    L.oracle_vecadd.argtypes = [dp, dp, C.c_double, C.c_size_t]
    L.oracle_vecdot.argtypes = [dp, dp, C.c_size_t]
    L.oracle_vecdot.restype = C.c_double
    L.oracle_vecscale.argtypes = [dp, C.c_double, C.c_size_t]
    L.oracle_veccpy.argtypes = [dp, dp, C.c_size_t]
    L.oracle_vecncpy.argtypes = [dp, dp, C.c_size_t]
    L.oracle_vecdiff.argtypes = [dp, dp, dp, C.c_size_t]
    L.oracle_vec2norm.argtypes = [dp, C.c_size_t]
    L.oracle_vec2norm.restype = C.c_double
    L.oracle_vec2norminv.argtypes = [dp, C.c_size_t]
    L.oracle_vec2norminv.restype = C.c_double
    L.oracle_signum.argtypes = [C.c_double]
    L.oracle_signum.restype = C.c_double
    L.oracle_x1norm.argtypes = [C.c_double, C.c_size_t, C.c_size_t, dp]
    L.oracle_x1norm.restype = C.c_double
    L.oracle_pseudo_gradient.argtypes = [C.c_double, C.c_size_t, C.c_size_t, dp, dp, dp, C.c_size_t]
    L.oracle_project.argtypes = [dp, dp, C.c_size_t, C.c_size_t, C.c_int]
    L.oracle_orthant_select.argtypes = [dp, dp, dp, C.c_size_t]
    L.oracle_two_loop.argtypes = [C.POINTER(dp), C.POINTER(dp), dp, dp, dp, C.c_double, C.c_size_t, C.c_size_t,
                                  C.c_size_t, C.c_size_t]
    L.oracle_two_loop.restype = C.c_size_t
    L.oracle_history_update.argtypes = [dp, dp, dp, dp, dp, dp, dp, dp, C.c_double, C.c_int, C.c_size_t, dp]
    L.oracle_build.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Param), dp, C.c_size_t, C.c_void_p, C.c_void_p]
    L.oracle_is_converged.argtypes = [C.c_void_p]
    L.oracle_propagate.argtypes = [C.c_void_p, C.POINTER(Progress)]
    L.oracle_get_report.argtypes = [C.c_void_p, C.POINTER(Report)]
    L.oracle_free.argtypes = [C.c_void_p]
    L.oracle_state_vec.argtypes = [C.c_void_p, C.c_int]
    L.oracle_state_vec.restype = dp
    L.oracle_state_hist.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.oracle_state_hist.restype = dp
    for nm in ("oracle_state_ys", "oracle_state_alpha"):
        getattr(L, nm).argtypes = [C.c_void_p, C.c_size_t]
        getattr(L, nm).restype = C.c_double
    for nm in ("oracle_state_end", "oracle_state_k"):
        getattr(L, nm).argtypes = [C.c_void_p]
        getattr(L, nm).restype = C.c_size_t
    for nm in ("oracle_state_step", "oracle_state_gamma"):
        getattr(L, nm).argtypes = [C.c_void_p]
        getattr(L, nm).restype = C.c_double
    L.oracle_state_ls_error.argtypes = [C.c_void_p]
    L.oracle_state_ls_error.restype = C.c_char_p
    L.oracle_state_ls_failed.argtypes = [C.c_void_p]
    L.oracle_line_search.argtypes = [C.c_void_p, dp, C.POINTER(C.c_uint64)]
    L.oracle_hash_u01.argtypes = [C.c_uint64, C.c_uint64]
    L.oracle_hash_u01.restype = C.c_double
    _lib = L
    return L


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


# --------------------------------------------------------------------------- primitives
def vecadd(y, x, c):
    lib().oracle_vecadd(_dp(y), _dp(x), c, min(len(x), len(y)))


def vecdot(x, y):
    return lib().oracle_vecdot(_dp(x), _dp(y), min(len(x), len(y)))


def vecscale(y, c):
    lib().oracle_vecscale(_dp(y), c, len(y))


def veccpy(y, x):
    lib().oracle_veccpy(_dp(y), _dp(x), min(len(x), len(y)))


def vecncpy(y, x):
    lib().oracle_vecncpy(_dp(y), _dp(x), min(len(x), len(y)))


def vecdiff(z, x, y):
    lib().oracle_vecdiff(_dp(z), _dp(x), _dp(y), min(len(x), len(y), len(z)))


def vec2norm(x):
    return lib().oracle_vec2norm(_dp(x), len(x))


def vec2norminv(x):
    return lib().oracle_vec2norminv(_dp(x), len(x))


def two_loop(s_list, y_list, ys, alpha, d, gamma, m, k, end):
    """lbfgs.rs:569-604 on numpy arrays; d and alpha are updated in place; returns new end."""
    dp = C.POINTER(C.c_double)
    S = (dp * m)(*[_dp(a) for a in s_list])
    Y = (dp * m)(*[_dp(a) for a in y_list])
    return lib().oracle_two_loop(S, Y, _dp(ys), _dp(alpha), _dp(d), gamma, m, k, end, len(d))


def history_update(s, y, x, xp, gx, gp, step, damping):
    """lbfgs.rs:640-692; returns (status, ys, gamma, aux[5])."""
    ys = np.zeros(1)
    gamma = np.zeros(1)
    aux = np.zeros(5)
    rc = lib().oracle_history_update(_dp(s), _dp(y), _dp(ys), _dp(gamma), _dp(x), _dp(xp), _dp(gx), _dp(gp),
                                     step, int(damping), len(x), _dp(aux))
    return rc, float(ys[0]), float(gamma[0]), aux


# --------------------------------------------------------------------------- objectives
SEED_QUAD_A, SEED_QUAD_B = 0x5EED0001, 0x5EED0002
SEED_LOGI_A, SEED_LOGI_T = 0x5EED0003, 0x5EED0004


@dataclass
class Builtin:
    """A C-side objective: (symbol name, user struct)."""
    name: str
    user: object = None


def rosenbrock():
    return Builtin("oracle_obj_rosenbrock")


def quadratic(global_offset=0):
    return Builtin("oracle_obj_quadratic", HashedObj(global_offset, SEED_QUAD_A, SEED_QUAD_B))


def logistic(global_offset=0):
    return Builtin("oracle_obj_logistic", HashedObj(global_offset, SEED_LOGI_A, SEED_LOGI_T))


def lj():
    return Builtin("oracle_obj_lj")


class LjNbr(C.Structure):
    _fields_ = [("nbr", C.POINTER(C.c_int32)), ("max_nbr", C.c_uint32), ("_pad", C.c_uint32), ("cutoff", C.c_double)]


def lj_neighbors(nbr_index, cutoff):
    """nbr_index: int32 array [max_nbr, natoms] (-1 = empty)."""
    tab = np.ascontiguousarray(nbr_index, dtype=np.int32)
    u = LjNbr(tab.ctypes.data_as(C.POINTER(C.c_int32)), tab.shape[0], 0, float(cutoff))
    b = Builtin("oracle_obj_lj_neighbors", u)
    b._keep = tab
    return b


class LjCells(C.Structure):
    _fields_ = [("cutoff", C.c_double)]


def lj_cells(cutoff):
    """E = sum_{i<j, r<rc} [v(r) - v(rc)] through a cell list rebuilt at every call (a function of x alone)."""
    return Builtin("oracle_obj_lj_cells", LjCells(float(cutoff)))


def _resolve_eval(evaluate):
    """Return (callable pointer as c_void_p, user pointer, keepalive)."""
    L = lib()
    if isinstance(evaluate, Builtin):
        fn = C.cast(getattr(L, evaluate.name), C.c_void_p)
        user = C.cast(C.pointer(evaluate.user), C.c_void_p) if evaluate.user is not None else None
        return fn, user, evaluate

    def tramp(_user, xp, gp, n, failed):
        x = np.ctypeslib.as_array(xp, shape=(n,))
        g = np.ctypeslib.as_array(gp, shape=(n,))
        try:
            return float(evaluate(x, g))
        except Exception:  # Err(..) in the reference
            failed[0] = 1
            return 0.0

    cb = EVAL_CB(tramp)
    return C.cast(cb, C.c_void_p), None, cb


def eval_builtin(obj: Builtin, x: np.ndarray):
    """Evaluate a C-side objective once: returns (f, g)."""
    L = lib()
    fn = getattr(L, obj.name)
    fn.restype = C.c_double
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_int)]
    g = np.zeros_like(x)
    failed = C.c_int(0)
    user = C.cast(C.pointer(obj.user), C.c_void_p) if obj.user is not None else None
    f = fn(user, _dp(x), _dp(g), len(x), C.byref(failed))
    return f, g


# --------------------------------------------------------------------------- builder (lbfgs.rs:185-384)
class Lbfgs:
    def __init__(self):
        self.param = Param()
        lib().oracle_param_default(C.byref(self.param))

    # setters keep the reference's names and assertions
    def with_epsilon(self, epsilon):
        assert not np.signbit(epsilon), "Invalid parameter epsilon specified."
        self.param.epsilon = epsilon
        return self

    def with_initial_step_size(self, b):
        assert not np.signbit(b)
        self.param.initial_inverse_hessian = b
        return self

    def with_max_step_size(self, s):
        assert not np.signbit(s)
        self.param.max_step_size = s
        return self

    def with_damping(self, damped):
        self.param.damping = int(bool(damped))
        return self

    def with_orthantwise(self, c, start, end=None):
        assert not np.signbit(c)
        self.param.orthantwise = 1
        self.param.owl_c = c
        self.param.owl_start = start
        self.param.owl_end = -1 if end is None else end
        return self

    def with_linesearch_ftol(self, ftol):
        assert ftol >= 0.0
        self.param.ftol = ftol
        return self

    def with_linesearch_gtol(self, gtol):
        assert 0.0 <= gtol < 1.0 and gtol > self.param.ftol
        self.param.gtol = gtol
        return self

    def with_gradient_only(self):
        self.param.gradient_only = 1
        self.param.damping = 1
        self.param.ls_algorithm = LS_BT_STRONGWOLFE
        return self

    def with_max_linesearch(self, n):
        self.param.max_linesearch = n
        return self

    def with_linesearch_xtol(self, xtol):
        assert xtol >= 0.0
        self.param.xtol = xtol
        return self

    def with_linesearch_min_step(self, min_step):
        assert min_step >= 0.0
        self.param.min_step = min_step
        return self

    def with_max_iterations(self, niter):
        self.param.max_iterations = niter
        return self

    def with_max_evaluations(self, neval):
        self.param.max_evaluations = neval
        return self

    def with_fx_delta(self, delta, past):
        assert delta >= 0.0
        self.param.past = past
        self.param.delta = delta
        return self

    def with_linesearch_algorithm(self, algo):
        table = {
            "MoreThuente": LS_MORETHUENTE,
            "BacktrackingArmijo": LS_BT_ARMIJO,
            "BacktrackingStrongWolfe": LS_BT_STRONGWOLFE,
            "BacktrackingWolfe": LS_BT_WOLFE,
            "Backtracking": LS_BT_WOLFE,
        }
        if algo not in table:
            raise NotImplementedError(algo)  # unimplemented!() lbfgs.rs:379
        self.param.ls_algorithm = table[algo]
        return self

    def with_m(self, m):
        """EXTENSION (not in the reference API): history length."""
        assert m >= 1
        self.param.m = m
        return self

    def build(self, x, evaluate):
        return State(self.param, x, evaluate)

    def minimize(self, x, evaluate, progress=None):
        """lbfgs.rs:399-421.  progress(p: dict) -> bool (True cancels)."""
        st = self.build(x, evaluate)
        try:
            while not st.is_converged():
                prgr = st.propagate()
                if progress is not None and progress(prgr):
                    break
            return st.report()
        finally:
            st.close()


def lbfgs():
    return Lbfgs()


class State:
    """LbfgsState (lbfgs.rs:425-566) over the C oracle."""

    def __init__(self, param, x, evaluate):
        assert x.dtype == np.float64 and x.flags["C_CONTIGUOUS"]
        self.x = x
        self.n = len(x)
        self.m = int(param.m)
        self._fn, self._user, self._keep = _resolve_eval(evaluate)
        self._h = C.c_void_p()
        p = Param()
        C.memmove(C.byref(p), C.byref(param), C.sizeof(Param))
        rc = lib().oracle_build(C.byref(self._h), C.byref(p), _dp(x), self.n, self._fn, self._user)
        if rc != 0:
            raise OracleError(rc)

    def close(self):
        if self._h:
            lib().oracle_free(self._h)
            self._h = C.c_void_p()

    def is_converged(self):
        return bool(lib().oracle_is_converged(self._h))

    def propagate(self):
        pr = Progress()
        rc = lib().oracle_propagate(self._h, C.byref(pr))
        if rc != 0:
            raise OracleError(rc)
        return dict(fx=pr.fx, xnorm=pr.xnorm, gnorm=pr.gnorm, step=pr.step, niter=pr.niter, neval=pr.neval,
                    ncall=pr.ncall, x=self.x, gx=self.vec("gx"))

    def report(self):
        r = Report()
        lib().oracle_get_report(self._h, C.byref(r))
        return dict(fx=r.fx, xnorm=r.xnorm, gnorm=r.gnorm, neval=r.neval)

    _VEC = {"gx": 0, "xp": 1, "gp": 2, "pg": 3, "wp": 4, "d": 5, "x": 6}

    def vec(self, name):
        p = lib().oracle_state_vec(self._h, self._VEC[name])
        return np.ctypeslib.as_array(p, shape=(self.n,))

    def hist(self, slot, which):
        p = lib().oracle_state_hist(self._h, slot, 1 if which == "y" else 0)
        return np.ctypeslib.as_array(p, shape=(self.n,))

    def ys(self, slot):
        return lib().oracle_state_ys(self._h, slot)

    def alpha(self, slot):
        return lib().oracle_state_alpha(self._h, slot)

    @property
    def end(self):
        return lib().oracle_state_end(self._h)

    @property
    def k(self):
        return lib().oracle_state_k(self._h)

    @property
    def step(self):
        return lib().oracle_state_step(self._h)

    @property
    def gamma(self):
        return lib().oracle_state_gamma(self._h)

    @property
    def ls_failed(self):
        return bool(lib().oracle_state_ls_failed(self._h))

    @property
    def ls_error(self):
        return lib().oracle_state_ls_error(self._h).decode()
