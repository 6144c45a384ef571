//! `liblbfgs` API (ybyygu/rust-lbfgs v0.2.0) on an MI355X: same builder, same `minimize` signature, every
//! n-vector resident in HBM.  UNTESTED -- written against the C headers in an image without a Rust toolchain;
//! see INTEGRATION.md.  The numerical behaviour is that of `liblbfgs_solver.so` (parity-tested against a CPU
//! restatement of the crate), not of this file: everything below is argument marshalling.
//!
//! ```ignore
//! use liblbfgs_hip::lbfgs;
//! let mut x = vec![0.0; 1_000_000];
//! let report = lbfgs()
//!     .with_max_iterations(100)
//!     .minimize(&mut x, |x, gx| { /* fill gx, return Ok(fx) */ Ok(0.0) }, |prgr| { println!("{}", prgr.fx); false })?;
//! ```
//! `minimize` keeps `E: FnMut(&[f64], &mut [f64]) -> Result<f64>` (src/lbfgs.rs:401): x is downloaded and gx
//! uploaded around every evaluation.  That is the compatibility path (PCIe-bound); `minimize_builtin` runs one of
//! the device-resident objectives with no host traffic, and a device-pointer closure can be passed through
//! `ffi::lbfgs_evaluator { kind: LBFGS_EVAL_DEVICE, .. }` by callers that own HIP kernels -- optionally with
//! `device_probe` / `device_accept`, so that a line search of T trials costs T probes (f and g.d, nothing written) and one
//! materialisation of the accepted point instead of T full evaluations (include/lbfgs_solver.h).
pub mod ffi;

use anyhow::{anyhow, bail, Result};
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::panic::{catch_unwind, AssertUnwindSafe};
use std::ptr;

/// Progress (src/core.rs:223-250).  `x` and `gx` are host copies made for the callback; they are empty when
/// `with_progress_vectors(false)` was requested (saves two PCIe transfers per iteration).
pub struct Progress<'a> {
    pub x: &'a [f64],
    pub gx: &'a [f64],
    pub fx: f64,
    pub xnorm: f64,
    pub gnorm: f64,
    pub step: f64,
    pub niter: usize,
    pub neval: usize,
    pub ncall: usize,
}

/// Report (src/core.rs:273-285)
#[derive(Debug, Clone)]
pub struct Report {
    pub fx: f64,
    pub xnorm: f64,
    pub gnorm: f64,
    pub neval: usize,
}

/// Device-resident objectives of `include/lbfgs_hip.h` (BASELINE.json's synthetic workloads + Rosenbrock).
#[derive(Clone, Copy)]
pub enum Builtin {
    Quadratic,
    Logistic,
    Rosenbrock,
}

/// `Lbfgs` (src/lbfgs.rs:181-383): the same `with_*` setters with the same assertions.
#[derive(Clone)]
pub struct Lbfgs {
    param: ffi::lbfgs_param,
    device: c_int,
    progress_vectors: bool,
}

/// src/lib.rs:74
pub fn lbfgs() -> Lbfgs {
    Lbfgs::default()
}

impl Default for Lbfgs {
    fn default() -> Self {
        let mut p = std::mem::MaybeUninit::<ffi::lbfgs_param>::zeroed();
        // lbfgs.rs:161-176, line.rs:151-162, orthantwise.rs:47-55
        let param = unsafe {
            ffi::lbfgs_param_default(p.as_mut_ptr());
            p.assume_init()
        };
        Lbfgs { param, device: 0, progress_vectors: true }
    }
}

impl Lbfgs {
    pub fn with_epsilon(mut self, epsilon: f64) -> Self {
        assert!(epsilon.is_sign_positive(), "Invalid parameter epsilon specified.");
        self.param.epsilon = epsilon;
        self
    }
    pub fn with_initial_step_size(mut self, b: f64) -> Self {
        assert!(b.is_sign_positive(), "Invalid beta parameter for scaling the initial step size.");
        self.param.initial_inverse_hessian = b;
        self
    }
    pub fn with_max_step_size(mut self, s: f64) -> Self {
        assert!(s.is_sign_positive(), "Invalid max_step_size parameter.");
        self.param.max_step_size = s;
        self
    }
    pub fn with_damping(mut self, damped: bool) -> Self {
        self.param.damping = damped as i32;
        self
    }
    pub fn with_orthantwise(mut self, c: f64, start: usize, end: impl Into<Option<usize>>) -> Self {
        assert!(c.is_sign_positive(), "Invalid parameter orthantwise c parameter specified.");
        self.param.orthantwise = 1;
        self.param.owl_c = c;
        self.param.owl_start = start as u64;
        self.param.owl_end = end.into().map(|e| e as i64).unwrap_or(-1);
        self
    }
    pub fn with_linesearch_ftol(mut self, ftol: f64) -> Self {
        assert!(ftol >= 0.0, "Invalid parameter ftol specified.");
        self.param.ftol = ftol;
        self
    }
    pub fn with_linesearch_gtol(mut self, gtol: f64) -> Self {
        assert!(gtol >= 0.0 && gtol < 1.0 && gtol > self.param.ftol, "Invalid parameter gtol specified.");
        self.param.gtol = gtol;
        self
    }
    pub fn with_gradient_only(mut self) -> Self {
        self.param.gradient_only = 1;
        self.param.damping = 1;
        self.param.ls_algorithm = ffi::LBFGS_LS_BACKTRACKING_STRONGWOLFE;
        self
    }
    pub fn with_max_linesearch(mut self, n: usize) -> Self {
        self.param.max_linesearch = n as u64;
        self
    }
    pub fn with_linesearch_xtol(mut self, xtol: f64) -> Self {
        assert!(xtol >= 0.0, "Invalid parameter xtol specified.");
        self.param.xtol = xtol;
        self
    }
    pub fn with_linesearch_min_step(mut self, min_step: f64) -> Self {
        assert!(min_step >= 0.0, "Invalid parameter min_step specified.");
        self.param.min_step = min_step;
        self
    }
    pub fn with_max_iterations(mut self, niter: usize) -> Self {
        self.param.max_iterations = niter as u64;
        self
    }
    pub fn with_max_evaluations(mut self, neval: usize) -> Self {
        self.param.max_evaluations = neval as u64;
        self
    }
    pub fn with_fx_delta(mut self, delta: f64, past: usize) -> Self {
        assert!(delta >= 0.0, "Invalid parameter delta specified.");
        self.param.past = past as u64;
        self.param.delta = delta;
        self
    }
    pub fn with_linesearch_algorithm(mut self, algo: &str) -> Self {
        self.param.ls_algorithm = match algo {
            "MoreThuente" => ffi::LBFGS_LS_MORETHUENTE,
            "BacktrackingArmijo" => ffi::LBFGS_LS_BACKTRACKING_ARMIJO,
            "BacktrackingStrongWolfe" => ffi::LBFGS_LS_BACKTRACKING_STRONGWOLFE,
            "BacktrackingWolfe" | "Backtracking" => ffi::LBFGS_LS_BACKTRACKING_WOLFE,
            _ => unimplemented!(), // src/lbfgs.rs:379
        };
        self
    }
    /// EXTENSION: number of corrections (the reference has no setter: always 6).
    pub fn with_m(mut self, m: usize) -> Self {
        assert!((1..=64).contains(&m), "m must be in 1..=64");
        self.param.m = m as u64;
        self
    }
    /// EXTENSION: HIP device index (default 0).
    pub fn with_device(mut self, device: i32) -> Self {
        self.device = device;
        self
    }
    /// EXTENSION: hand host copies of x and gx to the progress callback (default true, as the reference does).
    pub fn with_progress_vectors(mut self, on: bool) -> Self {
        self.progress_vectors = on;
        self
    }

    /// src/lbfgs.rs:399-421, same signature.  `x` is the start point and receives the result in place.
    pub fn minimize<E, G>(self, x: &mut [f64], mut eval_fn: E, prgr_fn: G) -> Result<Report>
    where
        E: FnMut(&[f64], &mut [f64]) -> Result<f64>,
        G: FnMut(&Progress) -> bool,
    {
        let mut bridge = HostBridge { f: &mut eval_fn, err: None }; // `&mut E` is itself FnMut
        let ev = host_evaluator(&mut bridge);
        let r = self.run(x, &ev, prgr_fn);
        match (r, bridge.err.take()) {
            (Err(_), Some(user_err)) => Err(user_err), // the closure's own Err, as lbfgs.rs:454 propagates it
            (r, _) => r,
        }
    }

    /// The same loop with a device-resident objective: nothing crosses PCIe except the progress copies.
    pub fn minimize_builtin<G>(self, x: &mut [f64], objective: Builtin, prgr_fn: G) -> Result<Report>
    where
        G: FnMut(&Progress) -> bool,
    {
        let mut obj = no_objective();
        match objective {
            Builtin::Quadratic => { obj.kind = ffi::LBFGS_HIP_OBJ_QUADRATIC; obj.seed_a = 0x5EED0001; obj.seed_b = 0x5EED0002; }
            Builtin::Logistic => { obj.kind = ffi::LBFGS_HIP_OBJ_LOGISTIC; obj.seed_a = 0x5EED0003; obj.seed_b = 0x5EED0004; }
            Builtin::Rosenbrock => { obj.kind = ffi::LBFGS_HIP_OBJ_ROSENBROCK; }
        }
        let ev = ffi::lbfgs_evaluator {
            kind: ffi::LBFGS_EVAL_BUILTIN,
            fuse_line_eval: 2,
            host: None,
            device: None,
            user: ptr::null_mut(),
            builtin: obj,
            device_probe: None,
            device_accept: None,
        };
        self.run(x, &ev, prgr_fn)
    }

    /// src/lbfgs.rs:443-481, same signature: caller-driven stepping through `LbfgsState`.
    pub fn build<'a, E>(self, x: &'a mut [f64], eval_fn: E) -> Result<LbfgsState<'a, E>>
    where
        E: FnMut(&[f64], &mut [f64]) -> Result<f64>,
    {
        let n = x.len();
        let mut bridge = Box::new(HostBridge { f: eval_fn, err: None }); // boxed: the C side keeps its address
        let ev = host_evaluator(&mut *bridge);
        let ctx = Context::new(self.device, n)?;
        let state = match State::build(&ctx, &self.param, x, &ev) {
            Ok(s) => s,
            Err(e) => return Err(bridge.err.take().unwrap_or(e)), // the closure's own Err (lbfgs.rs:454)
        };
        Ok(LbfgsState { state, ctx, bridge, x, gx: vec![0.0; n] })
    }

    fn run<G>(self, x: &mut [f64], ev: &ffi::lbfgs_evaluator, mut prgr_fn: G) -> Result<Report>
    where
        G: FnMut(&Progress) -> bool,
    {
        let n = x.len();
        let ctx = Context::new(self.device, n)?;
        let state = State::build(&ctx, &self.param, x, ev)?;
        let (mut hx, mut hg) = if self.progress_vectors { (vec![0.0; n], vec![0.0; n]) } else { (vec![], vec![]) };
        let result = (|| -> Result<Report> {
            loop {
                if state.is_converged()? {
                    break;
                }
                let p = state.propagate()?;
                if self.progress_vectors {
                    state.download(ffi::LBFGS_VEC_X, &mut hx)?;
                    state.download(ffi::LBFGS_VEC_GX, &mut hg)?;
                }
                let prgr = Progress {
                    x: &hx, gx: &hg, fx: p.fx, xnorm: p.xnorm, gnorm: p.gnorm, step: p.step,
                    niter: p.niter as usize, neval: p.neval as usize, ncall: p.ncall as usize,
                };
                if prgr_fn(&prgr) {
                    break; // cancelled (lbfgs.rs:412-416)
                }
            }
            state.report()
        })();
        // `x: &mut [f64]` is updated in place in the reference, also when the run ends with Err
        let copied = state.download(ffi::LBFGS_VEC_X, x);
        let report = result?;
        copied?;
        Ok(report)
    }
}

/// LbfgsState (src/lbfgs.rs:425-439).  `x` is kept current after every `propagate`, as in the reference where the
/// state holds `&mut x` (one download per iteration: this is the host-closure compatibility path).
pub struct LbfgsState<'a, E>
where
    E: FnMut(&[f64], &mut [f64]) -> Result<f64>,
{
    state: State,   // dropped first ...
    ctx: Context,   // ... then its context ...
    bridge: Box<HostBridge<E>>, // ... then the closure the C side pointed at
    x: &'a mut [f64],
    gx: Vec<f64>,
}

impl<'a, E> LbfgsState<'a, E>
where
    E: FnMut(&[f64], &mut [f64]) -> Result<f64>,
{
    /// src/lbfgs.rs:489-494
    pub fn is_converged(&self) -> bool {
        self.state.is_converged().expect("backend failure in is_converged")
    }

    /// src/lbfgs.rs:497-499
    pub fn report(&self) -> Report {
        self.state.report().expect("backend failure in report")
    }

    /// src/lbfgs.rs:503-560
    pub fn propagate(&mut self) -> Result<Progress> {
        let p = match self.state.propagate() {
            Ok(p) => p,
            Err(e) => {
                let _ = self.state.download(ffi::LBFGS_VEC_X, self.x);
                return Err(self.bridge.err.take().unwrap_or(e));
            }
        };
        self.state.download(ffi::LBFGS_VEC_X, self.x)?;
        self.state.download(ffi::LBFGS_VEC_GX, &mut self.gx)?;
        let _ = &self.ctx; // (kept alive by this struct)
        Ok(Progress {
            x: &*self.x, gx: &self.gx, fx: p.fx, xnorm: p.xnorm, gnorm: p.gnorm, step: p.step,
            niter: p.niter as usize, neval: p.neval as usize, ncall: p.ncall as usize,
        })
    }
}

fn no_objective() -> ffi::lbfgs_hip_objective {
    ffi::lbfgs_hip_objective { kind: 0, _pad: 0, seed_a: 0, seed_b: 0, nbr_index: ptr::null(), max_nbr: 0, _pad2: 0, cutoff: 0.0, skin: 0.0 }
}

// ---- the host closure behind the C callback ----------------------------------------------------------
struct HostBridge<F> {
    f: F,
    err: Option<anyhow::Error>,
}

/// The evaluator record for a closure that lives in `bridge` (whose address must stay valid while the state lives).
fn host_evaluator<F>(bridge: &mut HostBridge<F>) -> ffi::lbfgs_evaluator
where
    F: FnMut(&[f64], &mut [f64]) -> Result<f64>,
{
    ffi::lbfgs_evaluator {
        kind: ffi::LBFGS_EVAL_HOST,
        fuse_line_eval: 0,
        host: Some(host_trampoline::<F>),
        device: None,
        user: bridge as *mut HostBridge<F> as *mut c_void,
        builtin: no_objective(),
        device_probe: None,
        device_accept: None,
    }
}

unsafe extern "C" fn host_trampoline<F>(user: *mut c_void, x: *const f64, g: *mut f64, n_local: u64, failed: *mut c_int) -> f64
where
    F: FnMut(&[f64], &mut [f64]) -> Result<f64>,
{
    // never unwind across the FFI boundary: a panic in the closure is reported as a failed evaluation
    let outcome = catch_unwind(AssertUnwindSafe(|| {
        let bridge = &mut *(user as *mut HostBridge<F>);
        let xs = std::slice::from_raw_parts(x, n_local as usize);
        let gs = std::slice::from_raw_parts_mut(g, n_local as usize);
        match (bridge.f)(xs, gs) {
            Ok(fx) => Some(fx),
            Err(e) => {
                bridge.err = Some(e);
                None
            }
        }
    }));
    match outcome {
        Ok(Some(fx)) => fx,
        _ => {
            *failed = 1;
            0.0
        }
    }
}

// ---- RAII handles ---------------------------------------------------------------------------------
/// Device, stream, scalar board: what every `DeviceVec` of one problem lives on (`lbfgs_hip_ctx`).
pub struct Context {
    raw: *mut ffi::lbfgs_hip_ctx,
}

impl Context {
    pub fn new(device: c_int, n: usize) -> Result<Self> {
        // the #[repr(C)] mirrors in ffi.rs are passed by value: never talk to a library of another ABI version
        if unsafe { ffi::lbfgs_hip_abi_version() != ffi::LBFGS_HIP_ABI_VERSION } {
            bail!("liblbfgs_hip reports ABI version {}, this crate was written against {}",
                  unsafe { ffi::lbfgs_hip_abi_version() }, ffi::LBFGS_HIP_ABI_VERSION);
        }
        let mut raw = ptr::null_mut();
        let rc = unsafe { ffi::lbfgs_hip_ctx_create(&mut raw, device, n as u64, ptr::null(), ptr::null(), ptr::null_mut()) };
        if rc != ffi::LBFGS_HIP_OK {
            // there is no CPU fallback: no GPU => an error, never a silent slow path
            bail!("lbfgs_hip_ctx_create failed ({}): {}", rc, cstr(unsafe { ffi::lbfgs_hip_last_error(ptr::null()) }));
        }
        Ok(Context { raw })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { ffi::lbfgs_hip_ctx_destroy(self.raw) }
    }
}

// ---- the seam the crate itself names: `trait LbfgsMath` (src/math.rs:4-29) on device-resident vectors -----------------
// The reference implements it for `[f64]` (src/math.rs:31-82) and calls it from `Problem`, `lbfgs_two_loop_recursion`,
// `IterationData::update` and `propagate`.  liblbfgs_hip.so exports ONE symbol per method; this is the `impl` a port of the
// crate's own host code would sit on (the solver above takes the fused entry points instead: fewer passes, same results).
// UNCOMPILED like the rest of this file; tests/test_abi_exports.py checks every `ffi::` name against ffi.rs.
/// A device-resident n-vector (`Vec<f64>` of core.rs:24-39 kept in HBM).  Lives no longer than its `Context`.
pub struct DeviceVec<'c> {
    raw: *mut ffi::lbfgs_hip_vec,
    ctx: &'c Context,
    n: usize,
}

impl<'c> DeviceVec<'c> {
    /// vec![0.0; n]
    pub fn zeros(ctx: &'c Context, n: usize) -> Result<Self> {
        let mut raw = ptr::null_mut();
        let rc = unsafe { ffi::lbfgs_hip_vec_alloc(ctx.raw, &mut raw) };
        if rc != ffi::LBFGS_HIP_OK {
            bail!("lbfgs_hip_vec_alloc failed ({}): {}", rc, cstr(unsafe { ffi::lbfgs_hip_last_error(ctx.raw) }));
        }
        Ok(DeviceVec { raw, ctx, n })
    }
    pub fn from_slice(ctx: &'c Context, host: &[f64]) -> Result<Self> {
        let v = Self::zeros(ctx, host.len())?;
        v.ok(unsafe { ffi::lbfgs_hip_vec_upload(v.raw, host.as_ptr(), host.len() as u64) })?;
        Ok(v)
    }
    pub fn to_vec(&self) -> Result<Vec<f64>> {
        let mut host = vec![0.0; self.n];
        self.ok(unsafe { ffi::lbfgs_hip_vec_download(self.raw, host.as_mut_ptr(), self.n as u64) })?;
        Ok(host)
    }
    fn ok(&self, rc: c_int) -> Result<()> {
        if rc == ffi::LBFGS_HIP_OK { Ok(()) } else { Err(anyhow!("{} (status {})", cstr(unsafe { ffi::lbfgs_hip_last_error(self.ctx.raw) }), rc)) }
    }
}

impl<'c> Drop for DeviceVec<'c> {
    fn drop(&mut self) {
        unsafe { ffi::lbfgs_hip_vec_free(self.raw) }
    }
}

/// `trait LbfgsMath<T>` of src/math.rs:4-29 with the other operand device-resident too: same names, same meaning, same
/// arithmetic (y + (c*x) in two roundings; sums within 1e-10 of the reference's sequential ones).  A failing HIP call panics,
/// as an out-of-bounds slice would in the reference: the trait's methods return no `Result`.
pub trait LbfgsMath {
    /// y += c*x
    fn vecadd(&mut self, x: &Self, c: f64);
    /// s = x.dot(y)
    fn vecdot(&self, other: &Self) -> f64;
    /// y = x
    fn veccpy(&mut self, x: &Self);
    /// y = -x
    fn vecncpy(&mut self, x: &Self);
    /// z = x - y
    fn vecdiff(&mut self, x: &Self, y: &Self);
    /// y *= c
    fn vecscale(&mut self, c: f64);
    /// ||x||
    fn vec2norm(&self) -> f64;
    /// 1 / ||x||
    fn vec2norminv(&self) -> f64;
}

const MATH_SLOT: c_int = 250; // a board slot the solver does not use (include/lbfgs_hip.h: 256 slots)

impl<'c> LbfgsMath for DeviceVec<'c> {
    fn vecadd(&mut self, x: &Self, c: f64) {
        self.ok(unsafe { ffi::lbfgs_hip_vecadd(self.raw, x.raw, c) }).unwrap()
    }
    fn vecdot(&self, other: &Self) -> f64 {
        let mut s = 0.0;
        self.ok(unsafe { ffi::lbfgs_hip_vecdot(self.raw, other.raw, MATH_SLOT) }).unwrap();
        self.ok(unsafe { ffi::lbfgs_hip_scalars_read(self.ctx.raw, MATH_SLOT, 1, &mut s) }).unwrap();
        s
    }
    fn veccpy(&mut self, x: &Self) {
        self.ok(unsafe { ffi::lbfgs_hip_veccpy(self.raw, x.raw) }).unwrap()
    }
    fn vecncpy(&mut self, x: &Self) {
        self.ok(unsafe { ffi::lbfgs_hip_vecncpy(self.raw, x.raw) }).unwrap()
    }
    fn vecdiff(&mut self, x: &Self, y: &Self) {
        self.ok(unsafe { ffi::lbfgs_hip_vecdiff(self.raw, x.raw, y.raw) }).unwrap()
    }
    fn vecscale(&mut self, c: f64) {
        self.ok(unsafe { ffi::lbfgs_hip_vecscale(self.raw, c) }).unwrap()
    }
    fn vec2norm(&self) -> f64 {
        let mut s = 0.0;
        self.ok(unsafe { ffi::lbfgs_hip_vec2norm(self.raw, MATH_SLOT, &mut s) }).unwrap();
        s
    }
    fn vec2norminv(&self) -> f64 {
        let mut s = 0.0;
        self.ok(unsafe { ffi::lbfgs_hip_vec2norminv(self.raw, MATH_SLOT, &mut s) }).unwrap();
        s
    }
}

/// Must be dropped BEFORE the `Context` it was built on (locals: declare the context first; fields: declare the
/// state first -- Rust drops locals in reverse and fields in declaration order).
struct State {
    raw: *mut ffi::lbfgs_state,
}

impl State {
    fn build(ctx: &Context, param: &ffi::lbfgs_param, x0: &[f64], ev: &ffi::lbfgs_evaluator) -> Result<Self> {
        let mut raw = ptr::null_mut();
        let rc = unsafe { ffi::lbfgs_build(&mut raw, ctx.raw, param, x0.as_ptr(), ev) };
        if rc != ffi::LBFGS_HIP_OK {
            return Err(status_error(rc, cstr(unsafe { ffi::lbfgs_state_error(ptr::null()) })));
        }
        Ok(State { raw })
    }
    fn check(&self, rc: c_int) -> Result<()> {
        if rc == ffi::LBFGS_HIP_OK {
            Ok(())
        } else {
            Err(status_error(rc, cstr(unsafe { ffi::lbfgs_state_error(self.raw) })))
        }
    }
    fn is_converged(&self) -> Result<bool> {
        let mut c: c_int = 0;
        self.check(unsafe { ffi::lbfgs_is_converged(self.raw, &mut c) })?;
        Ok(c != 0)
    }
    fn propagate(&self) -> Result<ffi::lbfgs_progress> {
        let mut p = ffi::lbfgs_progress::default();
        self.check(unsafe { ffi::lbfgs_propagate(self.raw, &mut p) })?;
        Ok(p)
    }
    fn report(&self) -> Result<Report> {
        let mut r = ffi::lbfgs_report::default();
        self.check(unsafe { ffi::lbfgs_get_report(self.raw, &mut r) })?;
        Ok(Report { fx: r.fx, xnorm: r.xnorm, gnorm: r.gnorm, neval: r.neval as usize })
    }
    fn download(&self, which: c_int, host: &mut [f64]) -> Result<()> {
        self.check(unsafe { ffi::lbfgs_state_download(self.raw, which, host.as_mut_ptr()) })
    }
}

impl Drop for State {
    fn drop(&mut self) {
        unsafe { ffi::lbfgs_state_free(self.raw) }
    }
}

fn cstr(p: *const std::os::raw::c_char) -> String {
    if p.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

/// Status codes of include/lbfgs_solver.h.  The reference *panics* at two sites (orthantwise.rs:64, :160); the C
/// library reports them as codes and the panic is re-raised here so callers see the crate's behaviour.
fn status_error(rc: c_int, msg: String) -> anyhow::Error {
    if rc == ffi::LBFGS_PANIC_OWLQN_RANGE || rc == ffi::LBFGS_PANIC_ZERO_DIRECTION {
        panic!("{}", msg);
    }
    anyhow!("{} (status {})", msg, rc)
}
