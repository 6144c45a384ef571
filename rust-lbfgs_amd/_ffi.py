"""ctypes declarations of include/lbfgs_hip.h and include/lbfgs_solver.h.

The product loads ``liblbfgs_hip.so`` (HIP kernels + C-ABI) and ``liblbfgs_solver.so``
(host orchestration) from this directory and FAILS LOUDLY when they are missing or no
GPU is visible: there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))

BOARD_SLOTS = 256

# status codes (lbfgs_solver.h / lbfgs_hip.h)
OK = 0
ERR_EVALUATE, ERR_NEGATIVE_STEP, ERR_GRADONLY_MT = -1, -2, -3
ERR_X_NOT_CHANGED, ERR_GX_NOT_CHANGED, ERR_INVALID_DNORM = -4, -5, -6
PANIC_OWLQN_RANGE, PANIC_ZERO_DIRECTION, ERR_PARAM = -20, -21, -40
HIP_ERR_ARG, HIP_ERR_HIP, HIP_ERR_COMM, HIP_ERR_NOMEM, HIP_ERR_NO_DEVICE = -101, -102, -103, -104, -105

LS_MORETHUENTE, LS_BT_ARMIJO, LS_BT_STRONGWOLFE, LS_BT_WOLFE = 0, 1, 2, 3
EVAL_HOST, EVAL_DEVICE, EVAL_BUILTIN = 0, 1, 2
COMM_NONE, COMM_RCCL, COMM_CALLBACK, COMM_P2P = 0, 1, 2, 3
MAILBOX_AUTO, MAILBOX_DEVICE, MAILBOX_HOST = -1, 0, 1
ABI_VERSION = 5  # LBFGS_HIP_ABI_VERSION of the include/lbfgs_hip.h these declarations were written against
OBJ_QUADRATIC, OBJ_LOGISTIC, OBJ_ROSENBROCK, OBJ_LJ_ALLPAIRS, OBJ_LJ_NEIGHBORS, OBJ_LJ_CELLS = 1, 2, 3, 4, 5, 6
(K_TWOLOOP_STEP, K_TWOLOOP_EDGE, K_UPDATE, K_LINE, K_EVAL, K_OWLQN, K_BLAS1, K_COMM, K_TWOLOOP_ALL,
 K_TWOLOOP_RESIDENT) = range(10)
VEC_X, VEC_GX, VEC_XP, VEC_GP, VEC_PG, VEC_WP, VEC_D = range(7)
VEC_S0, VEC_Y0 = 100, 200


class Shard(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("n_global", C.c_uint64), ("offset", C.c_uint64),
                ("n_local", C.c_uint64)]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)


class Comm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("exclusive_device", C.c_int32), ("rccl_unique_id", C.c_void_p),
                ("callback", ALLREDUCE_CB), ("callback_user", C.c_void_p),
                ("p2p_mailbox", C.c_void_p), ("p2p_handles", C.c_void_p), ("p2p_timeout_s", C.c_double)]


XCHG_BINS = 48  # LBFGS_HIP_XCHG_BINS


class CommInfo(C.Structure):
    """lbfgs_hip_comm_info: what the communicator really spans and what its exchanges cost (lbfgs_hip_ctx_comm_info)."""
    _fields_ = [("kind", C.c_int32), ("world", C.c_int32), ("rank", C.c_int32), ("ranks_seen", C.c_int32),
                ("rank_seen", C.c_int32), ("mailbox_placement", C.c_int32), ("peers_device", C.c_int32),
                ("peers_host", C.c_int32), ("exclusive_device", C.c_int32), ("_pad", C.c_int32),
                ("two_loops", C.c_uint64), ("two_loop_exchanges", C.c_uint64), ("allreduce_launches", C.c_uint64),
                ("p2p_exchanges", C.c_uint64), ("resident_fallbacks", C.c_uint64),
                ("timed_exchanges", C.c_uint64 * 2), ("exchange_us", C.c_double * 2), ("local_wait_us", C.c_double * 2),
                ("exchange_us_max", C.c_double * 2), ("local_wait_us_max", C.c_double * 2), ("exchange_hist", (C.c_uint32 * XCHG_BINS) * 2)]


class Objective(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("seed_a", C.c_uint64), ("seed_b", C.c_uint64),
                ("nbr_index", C.c_void_p), ("max_nbr", C.c_uint32), ("_pad2", C.c_uint32), ("cutoff", C.c_double),
                ("skin", C.c_double)]


class Param(C.Structure):
    _fields_ = [
        ("m", C.c_uint64), ("epsilon", C.c_double), ("past", C.c_uint64), ("delta", C.c_double),
        ("max_iterations", C.c_uint64), ("max_evaluations", C.c_uint64),
        ("initial_inverse_hessian", C.c_double), ("max_step_size", C.c_double),
        ("damping", C.c_int32), ("constrain_step_size", C.c_int32),
        ("ls_algorithm", C.c_int32), ("gradient_only", C.c_int32),
        ("ftol", C.c_double), ("gtol", C.c_double), ("xtol", C.c_double),
        ("min_step", C.c_double), ("max_step", C.c_double), ("max_linesearch", C.c_uint64),
        ("orthantwise", C.c_int32), ("vector_free", C.c_int32), ("owl_c", C.c_double), ("owl_start", C.c_uint64),
        ("owl_end", C.c_int64),
    ]


HOST_EVAL_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint64,
                           C.POINTER(C.c_int))
DEVICE_EVAL_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                             C.POINTER(C.c_int))


# deferred trial points for device closures (lbfgs_solver.h, ABI version 4)
DEVICE_PROBE_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint64, C.c_void_p,
                              C.POINTER(C.c_double), C.POINTER(C.c_int))
DEVICE_ACCEPT_CB = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_uint64,
                               C.c_void_p, C.POINTER(C.c_int))


class Evaluator(C.Structure):
    _fields_ = [("kind", C.c_int32), ("fuse_line_eval", C.c_int32), ("host", HOST_EVAL_CB),
                ("device", DEVICE_EVAL_CB), ("user", C.c_void_p), ("builtin", Objective),
                ("device_probe", DEVICE_PROBE_CB), ("device_accept", DEVICE_ACCEPT_CB)]


class CProgress(C.Structure):
    _fields_ = [("fx", C.c_double), ("xnorm", C.c_double), ("gnorm", C.c_double), ("step", C.c_double),
                ("niter", C.c_uint64), ("neval", C.c_uint64), ("ncall", C.c_uint64)]


PROGRESS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(CProgress))


class CReport(C.Structure):
    _fields_ = [("fx", C.c_double), ("xnorm", C.c_double), ("gnorm", C.c_double), ("neval", C.c_uint64)]


# every symbol include/lbfgs_hip.h declares
HIP_SYMBOLS = """
lbfgs_hip_abi_version lbfgs_hip_build_id lbfgs_hip_device_count lbfgs_hip_rccl_unique_id lbfgs_hip_p2p_mailbox_create
lbfgs_hip_p2p_mailbox_create2 lbfgs_hip_p2p_mailbox_destroy lbfgs_hip_ctx_p2p_seal lbfgs_hip_ctx_create lbfgs_hip_ctx_destroy
lbfgs_hip_last_error lbfgs_hip_sync lbfgs_hip_stream lbfgs_hip_get_shard lbfgs_hip_set_grid lbfgs_hip_path_stats
lbfgs_hip_ctx_comm_info
lbfgs_hip_vec_alloc lbfgs_hip_vec_free lbfgs_hip_vec_upload lbfgs_hip_vec_download lbfgs_hip_vec_fill
lbfgs_hip_vec_ptr lbfgs_hip_vec_swap
lbfgs_hip_scalars_read lbfgs_hip_scalars_write lbfgs_hip_scalars_ptr lbfgs_hip_scalars_allreduce
lbfgs_hip_vecadd lbfgs_hip_vecadd_dev lbfgs_hip_vecdot lbfgs_hip_vecscale lbfgs_hip_veccpy lbfgs_hip_vecncpy
lbfgs_hip_vecdiff lbfgs_hip_vec2norm_sq lbfgs_hip_vec2norm lbfgs_hip_vec2norminv
lbfgs_hip_line_step lbfgs_hip_norms_sq
lbfgs_hip_history_create lbfgs_hip_history_destroy lbfgs_hip_history_s lbfgs_hip_history_y
lbfgs_hip_history_scalars_read lbfgs_hip_history_scalars_write lbfgs_hip_history_update lbfgs_hip_history_damp
lbfgs_hip_two_loop lbfgs_hip_two_loop_from lbfgs_hip_two_loop_owlqn lbfgs_hip_two_loop_gram lbfgs_hip_two_loop_unfused
lbfgs_hip_owlqn_post_eval lbfgs_hip_orthant_select lbfgs_hip_constrain_direction
lbfgs_hip_objective_eval lbfgs_hip_objective_line_eval lbfgs_hip_objective_owlqn_line_eval
lbfgs_hip_objective_owlqn_first_trial lbfgs_hip_objective_owlqn_trial_update
lbfgs_hip_objective_is_elementwise lbfgs_hip_objective_line_probe lbfgs_hip_history_update_from_step
lbfgs_hip_lj_cells_stats
lbfgs_hip_device_buffer_create
lbfgs_hip_device_buffer_destroy
lbfgs_hip_host_buffer_create lbfgs_hip_host_buffer_destroy
lbfgs_hip_prof_enable lbfgs_hip_prof_reset lbfgs_hip_prof_read
""".split()

SOLVER_SYMBOLS = """
lbfgs_param_default lbfgs_build lbfgs_is_converged lbfgs_propagate lbfgs_get_report lbfgs_state_free
lbfgs_state_error lbfgs_state_ls_error lbfgs_line_search lbfgs_state_download lbfgs_state_devptr lbfgs_state_info
lbfgs_state_history lbfgs_state_vec lbfgs_minimize lbfgs_solver_build_id lbfgs_state_vector_free_fallbacks
lbfgs_problem_new lbfgs_problem_evaluate lbfgs_problem_update_search_direction lbfgs_problem_dginit
lbfgs_problem_dg_unchecked lbfgs_problem_save_state lbfgs_problem_revert lbfgs_problem_take_line_step
lbfgs_problem_update_orthant_new_point lbfgs_problem_constrain_search_direction lbfgs_problem_norms
lbfgs_problem_status lbfgs_problem_set_linesearch
""".split()


def declare(L):
    """Attach argtypes/restypes for both headers to one CDLL that exports all symbols."""
    vp, dp, i, u64, dbl = C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_uint64, C.c_double
    sig = {
        "lbfgs_hip_abi_version": (i, []),
        "lbfgs_hip_build_id": (C.c_char_p, []),
        "lbfgs_solver_build_id": (C.c_char_p, []),
        "lbfgs_hip_p2p_mailbox_create2": (i, [i, i, C.POINTER(vp), vp]),
        "lbfgs_hip_ctx_p2p_seal": (i, [vp, C.POINTER(i)]),
        "lbfgs_hip_device_count": (i, [C.POINTER(i)]),
        "lbfgs_hip_rccl_unique_id": (i, [vp]),
        "lbfgs_hip_p2p_mailbox_create": (i, [i, C.POINTER(vp), vp]),
        "lbfgs_hip_p2p_mailbox_destroy": (None, [i, vp]),
        "lbfgs_hip_ctx_create": (i, [C.POINTER(vp), i, u64, C.POINTER(Shard), C.POINTER(Comm), vp]),
        "lbfgs_hip_ctx_destroy": (None, [vp]),
        "lbfgs_hip_last_error": (C.c_char_p, [vp]),
        "lbfgs_hip_sync": (i, [vp]),
        "lbfgs_hip_stream": (vp, [vp]),
        "lbfgs_hip_get_shard": (i, [vp, C.POINTER(Shard)]),
        "lbfgs_hip_set_grid": (i, [vp, i]),
        "lbfgs_hip_path_stats": (i, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "lbfgs_hip_ctx_comm_info": (i, [vp, C.POINTER(CommInfo)]),
        "lbfgs_hip_vec_alloc": (i, [vp, C.POINTER(vp)]),
        "lbfgs_hip_vec_free": (None, [vp]),
        "lbfgs_hip_vec_upload": (i, [vp, dp, u64]),
        "lbfgs_hip_vec_download": (i, [vp, dp, u64]),
        "lbfgs_hip_vec_fill": (i, [vp, dbl]),
        "lbfgs_hip_vec_ptr": (vp, [vp]),
        "lbfgs_hip_vec_swap": (i, [vp, vp]),
        "lbfgs_hip_scalars_read": (i, [vp, i, i, dp]),
        "lbfgs_hip_scalars_write": (i, [vp, i, i, dp]),
        "lbfgs_hip_scalars_ptr": (vp, [vp]),
        "lbfgs_hip_scalars_allreduce": (i, [vp, i, i]),
        "lbfgs_hip_vecadd": (i, [vp, vp, dbl]),
        "lbfgs_hip_vecadd_dev": (i, [vp, vp, i]),
        "lbfgs_hip_vecdot": (i, [vp, vp, i]),
        "lbfgs_hip_vecscale": (i, [vp, dbl]),
        "lbfgs_hip_veccpy": (i, [vp, vp]),
        "lbfgs_hip_vecncpy": (i, [vp, vp]),
        "lbfgs_hip_vecdiff": (i, [vp, vp, vp]),
        "lbfgs_hip_vec2norm_sq": (i, [vp, i]),
        "lbfgs_hip_vec2norm": (i, [vp, i, dp]),
        "lbfgs_hip_vec2norminv": (i, [vp, i, dp]),
        "lbfgs_hip_line_step": (i, [vp, vp, vp, dbl, vp, u64, u64]),
        "lbfgs_hip_norms_sq": (i, [vp, vp, i]),
        "lbfgs_hip_history_create": (i, [vp, i, C.POINTER(vp)]),
        "lbfgs_hip_history_destroy": (None, [vp]),
        "lbfgs_hip_history_s": (vp, [vp, i]),
        "lbfgs_hip_history_y": (vp, [vp, i]),
        "lbfgs_hip_history_scalars_read": (i, [vp, dp, dp]),
        "lbfgs_hip_history_scalars_write": (i, [vp, dp, dp]),
        "lbfgs_hip_history_update": (i, [vp, i, vp, vp, vp, vp, dbl, i, i]),
        "lbfgs_hip_history_damp": (i, [vp, i, vp, dbl, dbl]),
        "lbfgs_hip_two_loop": (i, [vp, vp, vp, u64, i, i, i, i, C.POINTER(i)]),
        "lbfgs_hip_two_loop_from": (i, [vp, vp, vp, u64, i, i, i, i, i, C.POINTER(i)]),
        "lbfgs_hip_two_loop_owlqn": (i, [vp, vp, vp, u64, i, i, i, i, u64, u64, C.POINTER(i)]),
        "lbfgs_hip_two_loop_gram": (i, [vp, vp, vp, u64, i, i, i, i, C.POINTER(i)]),
        "lbfgs_hip_two_loop_unfused": (i, [vp, vp, u64, i, i, i, C.POINTER(i)]),
        "lbfgs_hip_owlqn_post_eval": (i, [vp, vp, vp, dbl, u64, u64, i]),
        "lbfgs_hip_orthant_select": (i, [vp, vp, vp]),
        "lbfgs_hip_constrain_direction": (i, [vp, vp, u64, u64, i]),
        "lbfgs_hip_objective_eval": (i, [C.POINTER(Objective), vp, vp, i]),
        "lbfgs_hip_objective_line_eval": (i, [C.POINTER(Objective), vp, vp, vp, dbl, vp, i]),
        "lbfgs_hip_objective_is_elementwise": (i, [C.POINTER(Objective)]),
        "lbfgs_hip_lj_cells_stats": (i, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_uint32)]),
        "lbfgs_hip_objective_line_probe": (i, [C.POINTER(Objective), vp, vp, dbl, i]),
        "lbfgs_hip_history_update_from_step": (i, [vp, i, C.POINTER(Objective), vp, vp, vp, dbl, vp, vp, dbl, i, i]),
        "lbfgs_hip_objective_owlqn_line_eval": (i, [C.POINTER(Objective), vp, vp, vp, dbl, vp, vp, vp, dbl, u64, u64, i]),
        "lbfgs_hip_objective_owlqn_first_trial": (i, [C.POINTER(Objective), vp, vp, vp, dbl, vp, vp, vp, dbl, u64, u64, i]),
        "lbfgs_hip_objective_owlqn_trial_update": (i, [C.POINTER(Objective), vp, i, vp, vp, vp, dbl, vp, i, vp, vp, vp, dbl, u64, u64, i, i]),
        "lbfgs_hip_device_buffer_create": (i, [vp, vp, u64, C.POINTER(vp)]),
        "lbfgs_hip_host_buffer_create": (i, [vp, u64, C.POINTER(vp)]),
        "lbfgs_hip_host_buffer_destroy": (None, [vp, vp]),
        "lbfgs_hip_device_buffer_destroy": (None, [vp, vp]),
        "lbfgs_hip_prof_enable": (i, [vp, i]),
        "lbfgs_hip_prof_reset": (i, [vp]),
        "lbfgs_hip_prof_read": (i, [vp, i, C.POINTER(u64), dp]),
        # solver
        "lbfgs_param_default": (None, [C.POINTER(Param)]),
        "lbfgs_build": (i, [C.POINTER(vp), vp, C.POINTER(Param), dp, C.POINTER(Evaluator)]),
        "lbfgs_is_converged": (i, [vp, C.POINTER(i)]),
        "lbfgs_propagate": (i, [vp, C.POINTER(CProgress)]),
        "lbfgs_get_report": (i, [vp, C.POINTER(CReport)]),
        "lbfgs_state_free": (None, [vp]),
        "lbfgs_state_error": (C.c_char_p, [vp]),
        "lbfgs_state_ls_error": (C.c_char_p, [vp]),
        "lbfgs_line_search": (i, [vp, dp, C.POINTER(u64)]),
        "lbfgs_state_download": (i, [vp, i, dp]),
        "lbfgs_state_devptr": (vp, [vp, i]),
        "lbfgs_state_info": (i, [vp, C.POINTER(u64), C.POINTER(u64), dp, dp]),
        "lbfgs_state_vector_free_fallbacks": (i, [vp, C.POINTER(u64)]),
        "lbfgs_state_history": (vp, [vp]),
        "lbfgs_state_vec": (vp, [vp, i]),
        "lbfgs_problem_new": (i, [C.POINTER(vp), vp, C.POINTER(Param), dp, C.POINTER(Evaluator)]),
        "lbfgs_problem_evaluate": (i, [vp]),
        "lbfgs_problem_update_search_direction": (i, [vp]),
        "lbfgs_problem_dginit": (i, [vp, dp]),
        "lbfgs_problem_dg_unchecked": (i, [vp, dp]),
        "lbfgs_problem_save_state": (i, [vp]),
        "lbfgs_problem_revert": (i, [vp]),
        "lbfgs_problem_take_line_step": (i, [vp, dbl]),
        "lbfgs_problem_update_orthant_new_point": (i, [vp]),
        "lbfgs_problem_constrain_search_direction": (i, [vp]),
        "lbfgs_problem_norms": (i, [vp, dp, dp]),
        "lbfgs_problem_status": (i, [vp, dp, C.POINTER(u64), C.POINTER(i), C.POINTER(i)]),
        "lbfgs_problem_set_linesearch": (i, [vp, C.POINTER(Param)]),
        "lbfgs_minimize": (i, [vp, C.POINTER(Param), dp, C.POINTER(Evaluator), PROGRESS_CB, vp, C.POINTER(CReport),
                               C.c_char_p, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError = missing export: fail loudly
        fn.restype = res
        fn.argtypes = args
    return L


_LIB = None


def _one_hip_runtime():
    """PyTorch-ROCm bundles its own libamdhip64.so.7; if this library pulled in /opt/rocm's copy first and torch
    were imported later, the process would hold two HIP runtimes and the second sees no device.  So when torch
    is installed (and not yet imported) pre-load ITS runtime by path -- cheap, torch itself is not imported --
    and let liblbfgs_hip.so bind to it by SONAME.  LBFGS_HIP_RUNTIME=system keeps /opt/rocm's runtime."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("LBFGS_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    rt = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(rt):
        try:
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        except OSError:
            return
        # ... and if RCCL is needed later (world > 1) it must be the build that belongs to this runtime
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("LBFGS_HIP_RCCL", rccl)


def torch_before_rccl():
    """RCCL and PyTorch-ROCm share a stack of bundled libraries; dlopen'ing librccl before torch has initialised
    ends in a double free at interpreter exit (observed on ROCm 7.2 + torch 2.10+rocm7.0), while torch first is
    clean.  Every RCCL user of this package goes through torch.distributed for the rendezvous anyway, so make
    the order explicit."""
    import importlib.util
    import sys

    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load():
    """Load the product libraries.  Raises if they are not built (run rust_lbfgs_amd.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    libdir = os.environ.get("LBFGS_HIP_LIB_DIR", HERE)  # prebuilt libraries elsewhere (deployment, A/B builds)
    hip = os.path.join(libdir, "liblbfgs_hip.so")
    solver = os.path.join(libdir, "liblbfgs_solver.so")
    for p in (hip, solver):
        if not os.path.exists(p):
            raise ImportError(
                f"{p} is missing: the HIP extension is not built (python -c 'import rust_lbfgs_amd as r; r.build()'); "
                "this package has no CPU fallback")
    _one_hip_runtime()
    # liblbfgs_solver.so names liblbfgs_hip.so as a dependency (rpath $ORIGIN); RTLD_LOCAL keeps the
    # lbfgs_hip_* symbols out of the global namespace
    raw = C.CDLL(solver, mode=C.RTLD_LOCAL)
    # the ABI version FIRST, through a declaration of its own: an older library lacks symbols declare() asks for, and
    # "rebuild" is a better answer than an AttributeError about one of them
    try:
        raw.lbfgs_hip_abi_version.restype = C.c_int
        raw.lbfgs_hip_abi_version.argtypes = []
        got = raw.lbfgs_hip_abi_version()
    except AttributeError:
        got = None
    if got != ABI_VERSION:  # struct layouts are shared by value: never talk to a library with another one
        raise ImportError(f"{hip} has ABI version {got}, this package was written against {ABI_VERSION}: rebuild "
                          "(python -c 'import rust_lbfgs_amd as r; r.build()')")
    if os.path.realpath(libdir) == os.path.realpath(HERE) and os.environ.get("LBFGS_HIP_ALLOW_STALE") != "1":
        # In-tree libraries must belong to the checked-out sources (builds are content-addressed: _build.py).  A library of
        # the same ABI built from other sources -- older kernels -- is refused, not trusted; prebuilt libraries somewhere
        # else (LBFGS_HIP_LIB_DIR: deployments, A/B builds) carry whatever id they were built with and are taken as they are.
        # Checked BEFORE any symbol is declared, from the ids embedded in the FILES: a stale library of the same ABI version
        # may lack a symbol added since, and "rebuild" is the answer then too, not an AttributeError from declare().
        from . import _build

        have = (_build.embedded_id(hip, "LBFGS_HIP_BUILD_ID"), _build.embedded_id(solver, "LBFGS_SOLVER_BUILD_ID"))
        try:
            want = (_build.hip_build_id(), _build.solver_build_id())
        except OSError as e:  # (a deployment that ships the libraries without csrc/)
            raise ImportError(f"the sources the in-tree libraries are checked against are not available ({e}); set "
                              "LBFGS_HIP_ALLOW_STALE=1 to load them as they are, or LBFGS_HIP_LIB_DIR to name prebuilt ones") from e
        if have != want:
            raise ImportError(f"{hip} / {solver} carry build ids {have}, the checked-out sources hash to {want}: rebuild "
                              "(python -c 'import rust_lbfgs_amd as r; r.build()'), or set LBFGS_HIP_ALLOW_STALE=1 to load them anyway")
    lib = declare(raw)
    _LIB = lib
    return _LIB
