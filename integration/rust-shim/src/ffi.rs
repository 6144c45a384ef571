//! `extern "C"` mirror of `include/lbfgs_hip.h` and `include/lbfgs_solver.h` (the entry points this shim uses).
//! Field order and types of every `#[repr(C)]` struct follow the headers exactly.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct lbfgs_hip_ctx { _p: [u8; 0] }
#[repr(C)] pub struct lbfgs_hip_vec { _p: [u8; 0] }
#[repr(C)] pub struct lbfgs_hip_history { _p: [u8; 0] }
#[repr(C)] pub struct lbfgs_state { _p: [u8; 0] }

/// `LBFGS_HIP_ABI_VERSION` of the `include/lbfgs_hip.h` these declarations mirror; `Context::new` refuses a library
/// that reports another one (struct layouts are shared by value).
pub const LBFGS_HIP_ABI_VERSION: i32 = 5;
pub const LBFGS_HIP_OK: c_int = 0;
pub const LBFGS_ERR_EVALUATE: c_int = -1;
pub const LBFGS_PANIC_OWLQN_RANGE: c_int = -20;
pub const LBFGS_PANIC_ZERO_DIRECTION: c_int = -21;

pub const LBFGS_HIP_COMM_NONE: i32 = 0;
pub const LBFGS_HIP_COMM_RCCL: i32 = 1;
pub const LBFGS_HIP_COMM_CALLBACK: i32 = 2;
pub const LBFGS_HIP_COMM_P2P: i32 = 3;

pub const LBFGS_LS_MORETHUENTE: i32 = 0;
pub const LBFGS_LS_BACKTRACKING_ARMIJO: i32 = 1;
pub const LBFGS_LS_BACKTRACKING_STRONGWOLFE: i32 = 2;
pub const LBFGS_LS_BACKTRACKING_WOLFE: i32 = 3;

pub const LBFGS_EVAL_HOST: i32 = 0;
pub const LBFGS_EVAL_DEVICE: i32 = 1;
pub const LBFGS_EVAL_BUILTIN: i32 = 2;

pub const LBFGS_HIP_OBJ_QUADRATIC: i32 = 1;
pub const LBFGS_HIP_OBJ_LOGISTIC: i32 = 2;
pub const LBFGS_HIP_OBJ_ROSENBROCK: i32 = 3;
pub const LBFGS_HIP_OBJ_LJ_ALLPAIRS: i32 = 4;
pub const LBFGS_HIP_OBJ_LJ_NEIGHBORS: i32 = 5;

// lbfgs_state_download / lbfgs_state_vec selectors
pub const LBFGS_VEC_X: c_int = 0;
pub const LBFGS_VEC_GX: c_int = 1;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct lbfgs_hip_shard {
    pub rank: i32,
    pub world: i32,
    pub n_global: u64,
    pub offset: u64,
    pub n_local: u64,
}

pub type lbfgs_hip_allreduce_cb = Option<unsafe extern "C" fn(user: *mut c_void, buf: *mut f64, count: i32) -> c_int>;

#[repr(C)]
pub struct lbfgs_hip_comm {
    pub kind: i32,
    pub exclusive_device: i32,
    pub rccl_unique_id: *const c_void,
    pub callback: lbfgs_hip_allreduce_cb,
    pub callback_user: *mut c_void,
    pub p2p_mailbox: *mut c_void,
    pub p2p_handles: *const c_void,
    pub p2p_timeout_s: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct lbfgs_hip_objective {
    pub kind: i32,
    pub _pad: i32,
    pub seed_a: u64,
    pub seed_b: u64,
    pub nbr_index: *const c_void,
    pub max_nbr: u32,
    pub _pad2: u32,
    pub cutoff: f64,
    pub skin: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct lbfgs_param {
    pub m: u64,
    pub epsilon: f64,
    pub past: u64,
    pub delta: f64,
    pub max_iterations: u64,
    pub max_evaluations: u64,
    pub initial_inverse_hessian: f64,
    pub max_step_size: f64,
    pub damping: i32,
    pub constrain_step_size: i32,
    pub ls_algorithm: i32,
    pub gradient_only: i32,
    pub ftol: f64,
    pub gtol: f64,
    pub xtol: f64,
    pub min_step: f64,
    pub max_step: f64,
    pub max_linesearch: u64,
    pub orthantwise: i32,
    pub vector_free: i32,
    pub owl_c: f64,
    pub owl_start: u64,
    pub owl_end: i64,
}

pub type lbfgs_host_eval_cb =
    Option<unsafe extern "C" fn(user: *mut c_void, x: *const f64, g: *mut f64, n_local: u64, failed: *mut c_int) -> f64>;
pub type lbfgs_device_eval_cb = Option<
    unsafe extern "C" fn(user: *mut c_void, x_dev: *const c_void, g_dev: *mut c_void, n_local: u64, stream: *mut c_void,
                         failed: *mut c_int) -> f64,
>;

/// Deferred trial points for device closures (include/lbfgs_solver.h, ABI version 4): a line search reads f and g.d from each
/// trial (src/line.rs:283-288); `probe` returns f(xp + step*d) and stores grad.d in `dg_out` without writing a vector,
/// `accept` forms x = xp + step*d and g = grad f(x) once, for the accepted step.
pub type lbfgs_device_probe_cb = Option<
    unsafe extern "C" fn(user: *mut c_void, xp_dev: *const c_void, d_dev: *const c_void, step: f64, n_local: u64,
                         stream: *mut c_void, dg_out: *mut f64, failed: *mut c_int) -> f64,
>;
pub type lbfgs_device_accept_cb = Option<
    unsafe extern "C" fn(user: *mut c_void, xp_dev: *const c_void, d_dev: *const c_void, step: f64, x_out_dev: *mut c_void,
                         g_out_dev: *mut c_void, n_local: u64, stream: *mut c_void, failed: *mut c_int) -> f64,
>;

#[repr(C)]
pub struct lbfgs_evaluator {
    pub kind: i32,
    pub fuse_line_eval: i32,
    pub host: lbfgs_host_eval_cb,
    pub device: lbfgs_device_eval_cb,
    pub user: *mut c_void,
    pub builtin: lbfgs_hip_objective,
    pub device_probe: lbfgs_device_probe_cb,
    pub device_accept: lbfgs_device_accept_cb,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct lbfgs_progress {
    pub fx: f64,
    pub xnorm: f64,
    pub gnorm: f64,
    pub step: f64,
    pub niter: u64,
    pub neval: u64,
    pub ncall: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct lbfgs_report {
    pub fx: f64,
    pub xnorm: f64,
    pub gnorm: f64,
    pub neval: u64,
}

/// `lbfgs_hip_comm_info` (include/lbfgs_hip.h): what the communicator spans and what its exchanges cost.
pub const LBFGS_HIP_XCHG_BINS: usize = 48;
#[repr(C)]
#[derive(Clone, Copy)]
pub struct lbfgs_hip_comm_info {
    pub kind: i32,
    pub world: i32,
    pub rank: i32,
    pub ranks_seen: i32,
    pub rank_seen: i32,
    pub mailbox_placement: i32,
    pub peers_device: i32,
    pub peers_host: i32,
    pub exclusive_device: i32,
    pub _pad: i32,
    pub two_loops: u64,
    pub two_loop_exchanges: u64,
    pub allreduce_launches: u64,
    pub p2p_exchanges: u64,
    pub resident_fallbacks: u64,
    pub timed_exchanges: [u64; 2],
    pub exchange_us: [f64; 2],
    pub local_wait_us: [f64; 2],
    // ABI 5: the distribution behind those totals (bins: include/lbfgs_hip.h)
    pub exchange_us_max: [f64; 2],
    pub local_wait_us_max: [f64; 2],
    pub exchange_hist: [[u32; LBFGS_HIP_XCHG_BINS]; 2],
}

extern "C" {
    // ---- include/lbfgs_hip.h: context, vectors, scalar board --------------------------------------
    pub fn lbfgs_hip_abi_version() -> c_int;
    pub fn lbfgs_hip_build_id() -> *const c_char;
    pub fn lbfgs_hip_device_count(count: *mut c_int) -> c_int;
    pub fn lbfgs_hip_rccl_unique_id(out128: *mut c_void) -> c_int;
    pub fn lbfgs_hip_ctx_create(out: *mut *mut lbfgs_hip_ctx, device: c_int, n: u64, shard: *const lbfgs_hip_shard,
                                comm: *const lbfgs_hip_comm, stream: *mut c_void) -> c_int;
    pub fn lbfgs_hip_ctx_destroy(ctx: *mut lbfgs_hip_ctx);
    pub fn lbfgs_hip_last_error(ctx: *const lbfgs_hip_ctx) -> *const c_char;
    pub fn lbfgs_hip_sync(ctx: *mut lbfgs_hip_ctx) -> c_int;
    pub fn lbfgs_hip_vec_alloc(ctx: *mut lbfgs_hip_ctx, out: *mut *mut lbfgs_hip_vec) -> c_int;
    pub fn lbfgs_hip_vec_free(v: *mut lbfgs_hip_vec);
    pub fn lbfgs_hip_vec_upload(v: *mut lbfgs_hip_vec, host: *const f64, count: u64) -> c_int;
    pub fn lbfgs_hip_vec_download(v: *const lbfgs_hip_vec, host: *mut f64, count: u64) -> c_int;
    pub fn lbfgs_hip_vec_swap(a: *mut lbfgs_hip_vec, b: *mut lbfgs_hip_vec) -> c_int;
    pub fn lbfgs_hip_scalars_read(ctx: *mut lbfgs_hip_ctx, first: c_int, count: c_int, host: *mut f64) -> c_int;
    pub fn lbfgs_hip_scalars_write(ctx: *mut lbfgs_hip_ctx, first: c_int, count: c_int, host: *const f64) -> c_int;
    // ---- LbfgsMath (src/math.rs:31-82) on device vectors; reductions land on the scalar board -----
    pub fn lbfgs_hip_vecadd(y: *mut lbfgs_hip_vec, x: *const lbfgs_hip_vec, c: f64) -> c_int;
    pub fn lbfgs_hip_vecdot(x: *const lbfgs_hip_vec, y: *const lbfgs_hip_vec, out_slot: c_int) -> c_int;
    pub fn lbfgs_hip_vecscale(y: *mut lbfgs_hip_vec, c: f64) -> c_int;
    pub fn lbfgs_hip_veccpy(y: *mut lbfgs_hip_vec, x: *const lbfgs_hip_vec) -> c_int;
    pub fn lbfgs_hip_vecncpy(y: *mut lbfgs_hip_vec, x: *const lbfgs_hip_vec) -> c_int;
    pub fn lbfgs_hip_vecdiff(z: *mut lbfgs_hip_vec, x: *const lbfgs_hip_vec, y: *const lbfgs_hip_vec) -> c_int;
    pub fn lbfgs_hip_vec2norm_sq(x: *const lbfgs_hip_vec, out_slot: c_int) -> c_int;
    pub fn lbfgs_hip_path_stats(ctx: *mut lbfgs_hip_ctx, resident_two_loops: *mut u64, resident_elements: *mut u64) -> c_int;
    pub fn lbfgs_hip_ctx_comm_info(ctx: *mut lbfgs_hip_ctx, out: *mut lbfgs_hip_comm_info) -> c_int;
    pub fn lbfgs_hip_vec2norm(x: *const lbfgs_hip_vec, scratch_slot: c_int, out: *mut f64) -> c_int;
    pub fn lbfgs_hip_vec2norminv(x: *const lbfgs_hip_vec, scratch_slot: c_int, out: *mut f64) -> c_int;
    // ---- fused hot path ------------------------------------------------------------------------
    pub fn lbfgs_hip_line_step(x: *mut lbfgs_hip_vec, xp: *const lbfgs_hip_vec, d: *const lbfgs_hip_vec, step: f64,
                               wp: *const lbfgs_hip_vec, start: u64, end: u64) -> c_int;
    pub fn lbfgs_hip_norms_sq(x: *const lbfgs_hip_vec, g: *const lbfgs_hip_vec, out_slot: c_int) -> c_int;
    pub fn lbfgs_hip_history_create(ctx: *mut lbfgs_hip_ctx, m: c_int, out: *mut *mut lbfgs_hip_history) -> c_int;
    pub fn lbfgs_hip_history_destroy(h: *mut lbfgs_hip_history);
    pub fn lbfgs_hip_history_update(h: *mut lbfgs_hip_history, slot: c_int, x: *const lbfgs_hip_vec,
                                    xp: *const lbfgs_hip_vec, g: *const lbfgs_hip_vec, gp: *const lbfgs_hip_vec,
                                    step: f64, damping: c_int, out_slot: c_int) -> c_int;
    pub fn lbfgs_hip_history_damp(h: *mut lbfgs_hip_history, slot: c_int, gp: *const lbfgs_hip_vec, step: f64,
                                  theta: f64) -> c_int;
    pub fn lbfgs_hip_two_loop(h: *mut lbfgs_hip_history, d: *mut lbfgs_hip_vec, g: *const lbfgs_hip_vec, k: u64,
                              end: c_int, gamma_num_slot: c_int, gamma_den_slot: c_int, dnorm_slot: c_int,
                              new_end: *mut c_int) -> c_int;
    pub fn lbfgs_hip_two_loop_from(h: *mut lbfgs_hip_history, d: *mut lbfgs_hip_vec, g: *const lbfgs_hip_vec, k: u64,
                                   end: c_int, gamma_num_slot: c_int, gamma_den_slot: c_int, dnorm_slot: c_int,
                                   first_dot_slot: c_int, new_end: *mut c_int) -> c_int;
    pub fn lbfgs_hip_owlqn_post_eval(x: *const lbfgs_hip_vec, g: *const lbfgs_hip_vec, pg: *mut lbfgs_hip_vec, c: f64,
                                     start: u64, end: u64, out_slot: c_int) -> c_int;
    pub fn lbfgs_hip_orthant_select(wp: *mut lbfgs_hip_vec, xp: *const lbfgs_hip_vec, pg: *const lbfgs_hip_vec) -> c_int;
    pub fn lbfgs_hip_constrain_direction(d: *mut lbfgs_hip_vec, pg: *const lbfgs_hip_vec, start: u64, end: u64,
                                         out_slot: c_int) -> c_int;
    pub fn lbfgs_hip_objective_eval(obj: *const lbfgs_hip_objective, x: *const lbfgs_hip_vec, g: *mut lbfgs_hip_vec,
                                    out_slot: c_int) -> c_int;
    pub fn lbfgs_hip_objective_line_eval(obj: *const lbfgs_hip_objective, x: *mut lbfgs_hip_vec,
                                         xp: *const lbfgs_hip_vec, d: *const lbfgs_hip_vec, step: f64,
                                         g: *mut lbfgs_hip_vec, out_slot: c_int) -> c_int;

    // ---- include/lbfgs_solver.h: the optimiser loop over those entry points ---------------------------
    pub fn lbfgs_param_default(p: *mut lbfgs_param);
    pub fn lbfgs_build(out: *mut *mut lbfgs_state, ctx: *mut lbfgs_hip_ctx, param: *const lbfgs_param, x0: *const f64,
                       eval: *const lbfgs_evaluator) -> c_int;
    pub fn lbfgs_is_converged(st: *mut lbfgs_state, converged: *mut c_int) -> c_int;
    pub fn lbfgs_propagate(st: *mut lbfgs_state, out: *mut lbfgs_progress) -> c_int;
    pub fn lbfgs_get_report(st: *mut lbfgs_state, out: *mut lbfgs_report) -> c_int;
    pub fn lbfgs_state_free(st: *mut lbfgs_state);
    pub fn lbfgs_state_error(st: *const lbfgs_state) -> *const c_char;
    pub fn lbfgs_state_ls_error(st: *const lbfgs_state) -> *const c_char;
    pub fn lbfgs_state_download(st: *mut lbfgs_state, which: c_int, host: *mut f64) -> c_int;
}
