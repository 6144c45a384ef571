"""Host-side mirror of the reference's user API, over the native libraries.

    lbfgs().with_*(..).minimize(x, evaluate, progress) -> Report          (src/lbfgs.rs:399-421)
    lbfgs().build(x, evaluate) -> LbfgsState {is_converged, propagate, report}   (:443-565)

Names, argument meaning, defaults and error behaviour follow the reference crate
(`liblbfgs` v0.2.0); the work happens in liblbfgs_solver.so (host orchestration, C++)
and liblbfgs_hip.so (HIP kernels on MI355X).  Three kinds of `evaluate`:

  * a Python callable ``evaluate(x: np.ndarray, gx: np.ndarray) -> float`` -- the drop-in
    closure form ``FnMut(&[f64], &mut [f64]) -> Result<f64>`` on HOST slices (raise to
    signal ``Err``).  x/gx cross PCIe on every call: compatibility path;
  * ``DeviceEvaluate(fn)``: ``fn(x_ptr, g_ptr, n, stream) -> float`` on DEVICE pointers;
  * a built-in device-resident objective from ``objectives`` (benchmark workloads).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import Param


class LbfgsError(RuntimeError):
    """`Err(..)` of the reference's `anyhow::Result` (status code + message)."""

    def __init__(self, code, message=""):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


class LbfgsPanic(LbfgsError):
    """A `panic!`/`assert!` site of the reference (orthantwise.rs:64,160)."""


def _raise(code, message):
    if code in (_ffi.PANIC_OWLQN_RANGE, _ffi.PANIC_ZERO_DIRECTION):
        raise LbfgsPanic(code, message)
    raise LbfgsError(code, message)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ------------------------------------------------------------------------------------ context
def exchange_bin_edges_us():
    """[lo, hi) in microseconds of the bins of lbfgs_hip_comm_info.exchange_hist: quarter-microsecond bins up to 8 us, then
    octaves; the last bin is open (include/lbfgs_hip.h)."""
    edges = [(b / 4.0, (b + 1) / 4.0) for b in range(32)]
    edges += [(8.0 * 2 ** k, 8.0 * 2 ** (k + 1)) for k in range(_ffi.XCHG_BINS - 32)]
    edges[-1] = (edges[-1][0], float("inf"))
    return edges


def exchange_quantile(hist, q):
    """The q-quantile (0 < q <= 1) of the cross-rank exchange times from a histogram of lbfgs_hip_comm_info.exchange_hist
    (or a difference of two): linear inside the bin it falls into; the open last bin answers with its lower edge.
    -> microseconds, or None for an empty histogram."""
    total = sum(hist)
    if total <= 0:
        return None
    want, seen = q * total, 0
    for (lo, hi), c in zip(exchange_bin_edges_us(), hist):
        if c > 0 and seen + c >= want:
            return lo if hi == float("inf") else lo + (hi - lo) * (want - seen) / c
        seen += c
    return exchange_bin_edges_us()[-1][0]


class Context:
    """Device, stream, scalar board and communicator (lbfgs_hip_ctx)."""

    def __init__(self, n, device=0, shard=None, comm=None, stream=None):
        L = _ffi.load()
        self._L = L
        self._h = C.c_void_p()
        self._keep = comm  # keeps the callback / id buffer alive
        if comm is not None and comm.c.kind == _ffi.COMM_RCCL:
            _ffi.torch_before_rccl()
        shard_p = C.byref(shard) if shard is not None else None
        comm_p = C.byref(comm.c) if comm is not None else None
        rc = L.lbfgs_hip_ctx_create(C.byref(self._h), device, n, shard_p, comm_p, stream)
        if rc != 0:
            msg = L.lbfgs_hip_last_error(None).decode()
            self._h = C.c_void_p()
            raise LbfgsError(rc, msg)
        s = _ffi.Shard()
        L.lbfgs_hip_get_shard(self._h, C.byref(s))
        self.shard = s
        self.n_local = int(s.n_local)
        self.n_global = int(s.n_global)

    def check(self, rc):
        if rc != 0:
            _raise(rc, self._L.lbfgs_hip_last_error(self._h).decode())

    def close(self):
        if self._h:
            for buf in getattr(self, "_buffers", []):  # objective data uploaded for this context
                self._L.lbfgs_hip_device_buffer_destroy(self._h, buf)
            self._buffers = []
            self._L.lbfgs_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_buffer(self, array):
        """Raw device buffer holding `array` (objective data); freed with the context."""
        a = np.ascontiguousarray(array)
        buf = C.c_void_p()
        self.check(self._L.lbfgs_hip_device_buffer_create(self._h, a.ctypes.data_as(C.c_void_p), a.nbytes, C.byref(buf)))
        self.__dict__.setdefault("_buffers", []).append(buf)
        return buf

    # scalar board
    def scalars(self, first, count=1):
        out = np.zeros(count)
        self.check(self._L.lbfgs_hip_scalars_read(self._h, first, count, _dp(out)))
        return out

    def set_scalars(self, first, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self.check(self._L.lbfgs_hip_scalars_write(self._h, first, len(v), _dp(v)))

    def sync(self):
        self.check(self._L.lbfgs_hip_sync(self._h))

    def lj_cells_stats(self):
        """LJ_CELLS bookkeeping: (list builds, evaluations incl. re-evaluations, longest list of the latest build)."""
        r, e, m = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self.check(self._L.lbfgs_hip_lj_cells_stats(self._h, C.byref(r), C.byref(e), C.byref(m)))
        return r.value, e.value, m.value

    def resident_two_loops(self):
        """How many two-loop recursions of this context ran as the single on-chip-resident kernel."""
        r = C.c_uint64()
        self.check(self._L.lbfgs_hip_path_stats(self._h, C.byref(r), None))
        return r.value

    def resident_elements(self):
        """Elements of this rank's shard the last resident two-loop launch kept on the chip (n_local unless hybrid)."""
        e = C.c_uint64()
        self.check(self._L.lbfgs_hip_path_stats(self._h, None, C.byref(e)))
        return e.value

    def comm_info(self):
        """lbfgs_hip_ctx_comm_info as a dict: what the communicator really spans (`ranks_seen`: RCCL's own ncclCommCount, or
        the P2P mailboxes this rank mapped) and what its exchanges cost, measured on the device (class 0 = outside a
        two-loop, 1 = inside one).  Synchronises the stream."""
        ci = _ffi.CommInfo()
        self.check(self._L.lbfgs_hip_ctx_comm_info(self._h, C.byref(ci)))
        out = {k: getattr(ci, k) for k, _ in _ffi.CommInfo._fields_ if not k.startswith("_") and k not in (
            "timed_exchanges", "exchange_us", "local_wait_us", "exchange_us_max", "local_wait_us_max", "exchange_hist")}
        out["kind"] = {_ffi.COMM_NONE: "none", _ffi.COMM_RCCL: "rccl", _ffi.COMM_CALLBACK: "callback", _ffi.COMM_P2P: "p2p"}.get(ci.kind, ci.kind)
        out["mailbox_placement"] = {_ffi.MAILBOX_DEVICE: "device", _ffi.MAILBOX_HOST: "host"}.get(ci.mailbox_placement)
        if ci.kind == _ffi.COMM_CALLBACK:
            out["ranks_seen"] = None  # (the all-reduce is the caller's: the library sees no peer itself)
        for k in ("timed_exchanges", "exchange_us", "local_wait_us", "exchange_us_max", "local_wait_us_max"):
            out[k] = {"other": getattr(ci, k)[0], "two_loop": getattr(ci, k)[1]}
        # the distribution behind exchange_us (ABI 5): counts per duration bin, see exchange_quantile below
        out["exchange_hist"] = {"other": list(ci.exchange_hist[0]), "two_loop": list(ci.exchange_hist[1])}
        return out

    def set_grid(self, blocks):
        self.check(self._L.lbfgs_hip_set_grid(self._h, blocks))

    @property
    def stream(self):
        return self._L.lbfgs_hip_stream(self._h)

    # measurement
    def prof_enable(self, on=True):
        self.check(self._L.lbfgs_hip_prof_enable(self._h, int(on)))

    def prof_reset(self):
        self.check(self._L.lbfgs_hip_prof_reset(self._h))

    def prof_read(self, kclass):
        n = C.c_uint64()
        ms = C.c_double()
        self.check(self._L.lbfgs_hip_prof_read(self._h, kclass, C.byref(n), C.byref(ms)))
        return n.value, ms.value


# ------------------------------------------------------------------------------------ evaluators
class DeviceEvaluate:
    """`evaluate` on device pointers: fn(x_ptr:int, g_ptr:int, n_local:int, stream:int) -> float.

    Optional DEFERRED TRIAL POINTS (lbfgs_solver.h, lbfgs_evaluator.device_probe / device_accept): a line search reads only
    f and g.d from each trial (line.rs:283-288); with
        probe(xp_ptr, d_ptr, step, n_local, stream) -> (f(xp + step*d), grad f(xp + step*d) . d)       nothing written
        accept(xp_ptr, d_ptr, step, x_out_ptr, g_out_ptr, n_local, stream) -> None                     x = xp + step*d, g = grad f(x)
    a search of T trials costs T probes + one accept instead of T x (line step + evaluate + dot).  `accept` may be omitted
    (the solver then forms the accepted point with its line-step kernel + `fn`)."""

    def __init__(self, fn, probe=None, accept=None):
        self.fn = fn
        self.probe = probe
        self.accept = accept


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can view device memory it does not own."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchEvaluate(DeviceEvaluate):
    """`evaluate` written in PyTorch on the device-resident vectors (no PCIe traffic):

        def fn(x: torch.Tensor, g: torch.Tensor) -> float | torch.Tensor:   # x read-only, write g in place
            ...
    x and g are float64 CUDA(HIP) tensors that VIEW the optimiser's buffers of this rank's shard."""

    def __init__(self, fn, device=0, probe=None, accept=None):
        """probe(xp, d, step) -> (f, g.d) and accept(xp, d, step, x_out, g_out) on tensors that view the optimiser's buffers:
        the deferred-trial protocol of DeviceEvaluate."""
        import torch

        self._torch = torch
        self._dev = torch.device("cuda", device)
        view = lambda ptr, n: (torch.as_tensor(_DevArray(ptr, n), device=self._dev) if n else  # noqa: E731
                               torch.empty(0, dtype=torch.float64, device=self._dev))
        num = lambda v: float(v.detach()) if hasattr(v, "detach") else float(v)  # noqa: E731

        def raw_probe(xpptr, dptr, step, n, stream):
            f, dg = probe(view(xpptr, n), view(dptr, n), step)
            f, dg = num(f), num(dg)
            torch.cuda.synchronize(self._dev)
            return f, dg

        def raw_accept(xpptr, dptr, step, xptr, gptr, n, stream):
            accept(view(xpptr, n), view(dptr, n), step, view(xptr, n), view(gptr, n))
            torch.cuda.synchronize(self._dev)

        def raw(xptr, gptr, n, stream):
            if n == 0:
                return float(fn(torch.empty(0, dtype=torch.float64, device=self._dev),
                                torch.empty(0, dtype=torch.float64, device=self._dev)))
            xt = torch.as_tensor(_DevArray(xptr, n), device=self._dev)
            gt = torch.as_tensor(_DevArray(gptr, n), device=self._dev)
            f = fn(xt, gt)
            f = float(f.detach()) if hasattr(f, "detach") else float(f)  # .item() synchronises torch's stream: g is complete before the optimiser reads it
            torch.cuda.synchronize(self._dev)
            return f

        super().__init__(raw, probe=raw_probe if probe is not None else None,
                         accept=raw_accept if (probe is not None and accept is not None) else None)


@dataclass
class BuiltinObjective:
    kind: int
    seed_a: int = 0
    seed_b: int = 0
    fuse_line_eval: int = 1    # 0 / 1 / 2, see lbfgs_solver.h (bools accepted)
    nbr_index: object = None   # LJ_NEIGHBORS: int32 [max_nbr, natoms] host table (-1 = empty), uploaded per context
    cutoff: float = 0.0
    skin: float = 0.0          # LJ_CELLS: list radius = cutoff + skin
    max_nbr: int = 0           # LJ_CELLS: capacity of an atom's neighbour list (0 = the library's default, 128)

    def c_struct(self, ctx):
        """lbfgs_hip_objective for `ctx` (uploads the neighbour table once per context and keeps it alive)."""
        o = _ffi.Objective(self.kind, 0, self.seed_a, self.seed_b, None, int(self.max_nbr), 0, float(self.cutoff),
                           float(self.skin))
        if self.nbr_index is not None:
            cache = ctx.__dict__.setdefault("_objective_tables", {})  # one upload per (context, objective)
            if id(self) not in cache:
                tab = np.ascontiguousarray(self.nbr_index, dtype=np.int32)
                cache[id(self)] = (ctx.upload_buffer(tab), tab.shape[0], self)
            buf, max_nbr, _ = cache[id(self)]
            o.nbr_index, o.max_nbr = buf, max_nbr
        return o


def _make_evaluator(evaluate, ctx=None):
    """-> (Evaluator struct, keepalive, error-holder)"""
    ev = _ffi.Evaluator()
    holder = {"exc": None}
    if isinstance(evaluate, BuiltinObjective):
        ev.kind = _ffi.EVAL_BUILTIN
        ev.fuse_line_eval = int(evaluate.fuse_line_eval)
        ev.builtin = evaluate.c_struct(ctx)
        return ev, evaluate, holder
    if isinstance(evaluate, DeviceEvaluate):
        def dtramp(_user, xptr, gptr, n, stream, failed):
            try:
                return float(evaluate.fn(xptr, gptr, n, stream))
            except Exception as e:  # Err(..)
                holder["exc"] = e
                failed[0] = 1
                return 0.0

        cb = _ffi.DEVICE_EVAL_CB(dtramp)
        ev.kind = _ffi.EVAL_DEVICE
        ev.device = cb
        keep = [cb]
        if getattr(evaluate, "probe", None) is not None:
            def ptramp(_user, xpptr, dptr, step, n, stream, dg_out, failed):
                try:
                    f, dg = evaluate.probe(xpptr, dptr, step, n, stream)
                    dg_out[0] = float(dg)
                    return float(f)
                except Exception as e:  # Err(..)
                    holder["exc"] = e
                    failed[0] = 1
                    return 0.0

            pcb = _ffi.DEVICE_PROBE_CB(ptramp)
            ev.device_probe = pcb
            keep.append(pcb)
            if getattr(evaluate, "accept", None) is not None:
                def atramp(_user, xpptr, dptr, step, xptr, gptr, n, stream, failed):
                    try:
                        evaluate.accept(xpptr, dptr, step, xptr, gptr, n, stream)
                    except Exception as e:  # Err(..)
                        holder["exc"] = e
                        failed[0] = 1
                    return 0.0

                acb = _ffi.DEVICE_ACCEPT_CB(atramp)
                ev.device_accept = acb
                keep.append(acb)
        return ev, keep, holder

    def tramp(_user, xp, gp, n, failed):
        x = np.ctypeslib.as_array(xp, shape=(n,))
        g = np.ctypeslib.as_array(gp, shape=(n,))
        try:
            return float(evaluate(x, g))
        except Exception as e:  # Err(..)
            holder["exc"] = e
            failed[0] = 1
            return 0.0

    cb = _ffi.HOST_EVAL_CB(tramp)
    ev.kind = _ffi.EVAL_HOST
    ev.host = cb
    return ev, cb, holder


# ------------------------------------------------------------------------------------ Progress / Report
class Progress:
    """core.rs:223-250.  `x` and `gx` live on the device and are fetched on first access."""

    def __init__(self, state, c):
        self._state = state
        self.fx, self.xnorm, self.gnorm, self.step = c.fx, c.xnorm, c.gnorm, c.step
        self.niter, self.neval, self.ncall = c.niter, c.neval, c.ncall

    @property
    def x(self):
        return self._state.download("x")

    @property
    def gx(self):
        return self._state.download("gx")

    def __repr__(self):
        return (f"Progress(niter={self.niter}, neval={self.neval}, ncall={self.ncall}, fx={self.fx!r}, "
                f"xnorm={self.xnorm!r}, gnorm={self.gnorm!r}, step={self.step!r})")


@dataclass
class Report:
    """core.rs:273-285"""
    fx: float
    xnorm: float
    gnorm: float
    neval: int


# ------------------------------------------------------------------------------------ LbfgsState
class LbfgsState:
    """lbfgs.rs:425-566: caller-driven stepping."""

    _VEC = {"x": _ffi.VEC_X, "gx": _ffi.VEC_GX, "xp": _ffi.VEC_XP, "gp": _ffi.VEC_GP, "pg": _ffi.VEC_PG,
            "wp": _ffi.VEC_WP, "d": _ffi.VEC_D}

    def __init__(self, param, x, evaluate, ctx=None, device=0):
        L = _ffi.load()
        self._L = L
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._own_ctx = ctx is None
        self.ctx = ctx if ctx is not None else Context(len(x), device=device)
        if len(x) != self.ctx.n_local:
            raise LbfgsError(_ffi.ERR_PARAM, f"x has {len(x)} elements, the context's shard {self.ctx.n_local}")
        self.m = int(param.m)
        self._ev, self._keep, self._holder = _make_evaluator(evaluate, self.ctx)
        self._h = C.c_void_p()
        p = Param()
        C.memmove(C.byref(p), C.byref(param), C.sizeof(Param))
        rc = L.lbfgs_build(C.byref(self._h), self.ctx._h, C.byref(p), _dp(x), C.byref(self._ev))
        if rc != 0:
            msg = L.lbfgs_state_error(None).decode()
            self._h = C.c_void_p()
            if self._own_ctx:
                self.ctx.close()
            self._reraise(rc, msg)

    def _reraise(self, rc, msg):
        exc = self._holder.get("exc")
        if rc == _ffi.ERR_EVALUATE and exc is not None:
            self._holder["exc"] = None
            raise LbfgsError(rc, f"{msg}: {exc!r}") from exc
        _raise(rc, msg)

    def _check(self, rc):
        if rc != 0:
            self._reraise(rc, self._L.lbfgs_state_error(self._h).decode())

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

    def is_converged(self):
        c = C.c_int()
        self._check(self._L.lbfgs_is_converged(self._h, C.byref(c)))
        return bool(c.value)

    def propagate(self):
        p = _ffi.CProgress()
        self._check(self._L.lbfgs_propagate(self._h, C.byref(p)))
        return Progress(self, p)

    def report(self):
        r = _ffi.CReport()
        self._check(self._L.lbfgs_get_report(self._h, C.byref(r)))
        return Report(r.fx, r.xnorm, r.gnorm, r.neval)

    def line_search(self, step):
        """LineSearch::find (line.rs:193-223) on the current point; returns (ncall, step)."""
        s = C.c_double(step)
        n = C.c_uint64()
        self._check(self._L.lbfgs_line_search(self._h, C.byref(s), C.byref(n)))
        return n.value, s.value

    # --- introspection (parity tests, Progress.x / .gx)
    def download(self, name):
        if isinstance(name, str):
            which = self._VEC[name] if name in self._VEC else None
            if which is None:
                kind, slot = name[0], int(name[1:])  # "s3", "y0"
                which = (_ffi.VEC_S0 if kind == "s" else _ffi.VEC_Y0) + slot
        else:
            which = name
        out = np.zeros(self.ctx.n_local)
        self._check(self._L.lbfgs_state_download(self._h, which, _dp(out)))
        return out

    def devptr(self, name):
        return self._L.lbfgs_state_devptr(self._h, self._VEC[name])

    def info(self):
        k, e = C.c_uint64(), C.c_uint64()
        s, g = C.c_double(), C.c_double()
        self._check(self._L.lbfgs_state_info(self._h, C.byref(k), C.byref(e), C.byref(s), C.byref(g)))
        return dict(k=k.value, end=e.value, step=s.value, gamma=g.value)

    def vector_free_fallbacks(self):
        """EXTENSION (with_vector_free): iterations so far whose coefficient-space direction failed its run-time ||d||^2 check
        and was formed again by the exact recursion."""
        c = C.c_uint64()
        self._check(self._L.lbfgs_state_vector_free_fallbacks(self._h, C.byref(c)))
        return c.value

    def history_scalars(self):
        ys, al = np.zeros(self.m), np.zeros(self.m)
        h = self._L.lbfgs_state_history(self._h)
        self.ctx.check(self._L.lbfgs_hip_history_scalars_read(h, _dp(ys), _dp(al)))
        return ys, al

    @property
    def ls_error(self):
        return self._L.lbfgs_state_ls_error(self._h).decode()


# ------------------------------------------------------------------------------------ builder
def _assert(cond, msg):
    if not cond:
        raise AssertionError(msg)  # the reference's assert! panics (lbfgs.rs:195-361)


class Lbfgs:
    """lbfgs.rs:180-384.  Every setter keeps the reference's name, meaning and assertion."""

    _ALGOS = {
        "MoreThuente": _ffi.LS_MORETHUENTE,
        "BacktrackingArmijo": _ffi.LS_BT_ARMIJO,
        "BacktrackingStrongWolfe": _ffi.LS_BT_STRONGWOLFE,
        "BacktrackingWolfe": _ffi.LS_BT_WOLFE,
        "Backtracking": _ffi.LS_BT_WOLFE,
    }

    def __init__(self):
        self.param = Param()
        _ffi.load().lbfgs_param_default(C.byref(self.param))

    def with_epsilon(self, epsilon):
        _assert(not np.signbit(epsilon), "Invalid parameter epsilon specified.")
        self.param.epsilon = epsilon
        return self

    def with_initial_step_size(self, b):
        _assert(not np.signbit(b), "Invalid beta parameter for scaling the initial step size.")
        self.param.initial_inverse_hessian = b
        return self

    def with_max_step_size(self, s):
        _assert(not np.signbit(s), "Invalid max_step_size parameter.")
        self.param.max_step_size = s
        return self

    def with_damping(self, damped):
        self.param.damping = int(bool(damped))
        return self

    def with_orthantwise(self, c, start, end=None):
        _assert(not np.signbit(c), "Invalid parameter orthantwise c parameter specified.")
        self.param.orthantwise = 1
        self.param.owl_c = c
        self.param.owl_start = start
        self.param.owl_end = -1 if end is None else end
        return self

    def with_linesearch_ftol(self, ftol):
        _assert(ftol >= 0.0, "Invalid parameter ftol specified.")
        self.param.ftol = ftol
        return self

    def with_linesearch_gtol(self, gtol):
        _assert(0.0 <= gtol < 1.0 and gtol > self.param.ftol, "Invalid parameter gtol specified.")
        self.param.gtol = gtol
        return self

    def with_gradient_only(self):
        self.param.gradient_only = 1
        self.param.damping = 1
        self.param.ls_algorithm = _ffi.LS_BT_STRONGWOLFE
        return self

    def with_max_linesearch(self, n):
        self.param.max_linesearch = n
        return self

    def with_linesearch_xtol(self, xtol):
        _assert(xtol >= 0.0, "Invalid parameter xtol specified.")
        self.param.xtol = xtol
        return self

    def with_linesearch_min_step(self, min_step):
        _assert(min_step >= 0.0, "Invalid parameter min_step specified.")
        self.param.min_step = min_step
        return self

    def with_max_iterations(self, niter):
        self.param.max_iterations = niter
        return self

    def with_max_evaluations(self, neval):
        self.param.max_evaluations = neval
        return self

    def with_fx_delta(self, delta, past):
        _assert(delta >= 0.0, "Invalid parameter delta specified.")
        self.param.past = past
        self.param.delta = delta
        return self

    def with_linesearch_algorithm(self, algo):
        if algo not in self._ALGOS:
            raise NotImplementedError(algo)  # unimplemented!() lbfgs.rs:379
        self.param.ls_algorithm = self._ALGOS[algo]
        return self

    def with_m(self, m):
        """EXTENSION: number of corrections.  The reference has no setter (always 6);
        BASELINE.json's configs need m = 7 and m = 10."""
        _assert(1 <= m <= 64, "m must be in 1..=64")
        self.param.m = m
        return self

    def with_vector_free(self, on=True):
        """EXTENSION: run the two-loop recursion in Gram-coefficient space -- 4m+3 vector passes
        and 2 all-reduces per iteration instead of 8m+1 and 2m+1.  Same direction up to rounding (<= 1e-10
        relative in the parity tests); the exact recursion stays the default."""
        _assert(not on or self.param.m <= 10, "vector-free two-loop supports m <= 10")
        self.param.vector_free = int(bool(on))
        return self

    def build(self, x, evaluate, *, ctx=None, device=0):
        """lbfgs.rs:443-481"""
        return LbfgsState(self.param, x, evaluate, ctx=ctx, device=device)

    def minimize(self, x, evaluate, progress=None, *, ctx=None, device=0):
        """lbfgs.rs:399-421.  `x` (numpy f64) is the start point and receives the result in place.
        `progress(prgr) -> bool`: returning True cancels."""
        if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags["C_CONTIGUOUS"]):
            raise TypeError("x must be a C-contiguous float64 numpy array (it is updated in place)")
        state = self.build(x, evaluate, ctx=ctx, device=device)
        try:
            try:
                while not state.is_converged():
                    prgr = state.propagate()
                    if progress is not None and progress(prgr):
                        break  # "The minimization process has been canceled."
                return state.report()
            finally:
                x[:] = state.download("x")  # `x: &mut [f64]` always holds the current point
        finally:
            state.close()


def lbfgs():
    """lib.rs:74"""
    return Lbfgs()


def default_progress():
    """lib.rs:102-112"""

    def prgr_fn(prgr):
        print(f"Iteration {prgr.niter}, Evaluation {prgr.neval}:")
        # Rust `{:-12.6}`: width 12, 6 decimals, right-aligned (the `-` flag is unused)
        print(f" fx = {prgr.fx:12.6f} xnorm = {prgr.xnorm:12.6f}, gnorm = {prgr.gnorm:12.6f}, "
              f"ls = {prgr.ncall}, step = {prgr.step}")
        return False

    return prgr_fn


def default_evaluate():
    """lib.rs:79-94: Rosenbrock, as a host closure."""

    def evaluate(arr_x, gx):
        x0, x1 = arr_x[0::2], arr_x[1::2]
        t1 = 1.0 - x0
        t2 = 10.0 * (x1 - x0 * x0)
        gx[1::2] = 20.0 * t2
        gx[0::2] = -2.0 * (x0 * gx[1::2] + t1)
        fx = 0.0
        for v in (t1 * t1 + t2 * t2):
            fx += v
        return fx

    return evaluate
