// include/lbfgs.hpp -- the crate's public API (ybyygu/rust-lbfgs, `liblbfgs` v0.2.0) for C++ callers, header-only, over the
// C-ABI of lbfgs_solver.h / lbfgs_hip.h.
//
// The reference's host language is Rust; this image has no Rust toolchain, so next to the (uncompiled) shim crate in
// integration/rust-shim this header is the COMPILED mirror of the same surface -- same names, same argument meaning, same
// defaults and the same error behaviour -- so that a test written against the crate reads the same here:
//
//     std::vector<double> x = ...;                                   // let mut x = [...];
//     auto prb = liblbfgs::lbfgs()                                   // let prb = lbfgs()
//                    .with_orthantwise(1.0, 0, 99)                   //     .with_orthantwise(1.0, 0, 99)
//                    .minimize(x, evaluate, progress);               //     .minimize(&mut x, evaluate, progress)?;
//
//   lbfgs() -> Lbfgs                                  src/lib.rs:74,  src/lbfgs.rs:185-192
//   Lbfgs::with_*                                     src/lbfgs.rs:194-383  (every assert! of a setter throws liblbfgs::Panic)
//   Lbfgs::minimize(x, eval_fn, prgr_fn) -> Report    src/lbfgs.rs:399-421  (E: double(const double* x, double* gx, size_t n),
//                                                     an exception thrown by it is the closure's Err; G: bool(const Progress&),
//                                                     true cancels)
//   Lbfgs::build(x, eval_fn) -> LbfgsState            src/lbfgs.rs:443-481; is_converged / propagate / report :489-565
//   Progress{x, gx, fx, xnorm, gnorm, step, niter, neval, ncall}     src/core.rs:223-250 (x and gx are fetched from the device
//                                                     when asked for)
//   Report{fx, xnorm, gnorm, neval}                   src/core.rs:273-285
//   default_evaluate() / default_progress()           src/lib.rs:79-94, :101-114
// Result::Err comes back as liblbfgs::Error (code() = the status of lbfgs_solver.h), the crate's two panic! sites
// (orthantwise.rs:64,160) and the setters' assert!s as liblbfgs::Panic.  EXTENSIONS, marked as such: with_m (the reference has
// no setter for m), with_device, with_vector_free, minimize_builtin / device evaluators (x and g never leave HBM).
#ifndef LBFGS_HPP
#define LBFGS_HPP

#include <cmath>
#include <cstdio>
#include <exception>
#include <functional>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "lbfgs_solver.h"

namespace liblbfgs {

class Error : public std::runtime_error {  // anyhow::Error of a Result::Err
    int code_;

  public:
    Error(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }
};
class Panic : public Error {  // panic! / assert! of the reference
  public:
    using Error::Error;
};

struct Report {  // core.rs:273-285
    double fx = 0.0, xnorm = 0.0, gnorm = 0.0;
    std::size_t neval = 0;
};

class Progress {  // core.rs:223-250
    lbfgs_state* st_;
    std::size_t n_;
    mutable std::vector<double> x_, gx_;

  public:
    double fx, xnorm, gnorm, step;
    std::size_t niter, neval, ncall;
    Progress(lbfgs_state* st, std::size_t n, const lbfgs_progress& p)
        : st_(st), n_(n), fx(p.fx), xnorm(p.xnorm), gnorm(p.gnorm), step(p.step), niter(p.niter), neval(p.neval), ncall(p.ncall) {}
    const std::vector<double>& x() const {  // (the vectors live in HBM: downloaded on first use)
        if (x_.empty() && n_) {
            x_.resize(n_);
            if (lbfgs_state_download(st_, LBFGS_VEC_X, x_.data()) != LBFGS_OK) throw Error(LBFGS_HIP_ERR_HIP, lbfgs_state_error(st_));
        }
        return x_;
    }
    const std::vector<double>& gx() const {
        if (gx_.empty() && n_) {
            gx_.resize(n_);
            if (lbfgs_state_download(st_, LBFGS_VEC_GX, gx_.data()) != LBFGS_OK) throw Error(LBFGS_HIP_ERR_HIP, lbfgs_state_error(st_));
        }
        return gx_;
    }
};

namespace detail {
[[noreturn]] inline void raise(int rc, const std::string& msg) {
    if (rc == LBFGS_PANIC_OWLQN_RANGE || rc == LBFGS_PANIC_ZERO_DIRECTION || rc == LBFGS_ERR_PARAM) throw Panic(rc, msg);
    throw Error(rc, msg);
}
inline void require(bool cond, const char* msg) {  // assert!(cond, msg) of a setter
    if (!cond) throw Panic(LBFGS_ERR_PARAM, msg);
}
// the closure behind the C callback: exceptions never cross the C boundary
template <class E>
struct HostBridge {
    E* eval;
    std::exception_ptr err;
    static double call(void* user, const double* x, double* g, uint64_t n, int* failed) {
        auto* self = static_cast<HostBridge*>(user);
        try {
            return (*self->eval)(x, g, static_cast<std::size_t>(n));
        } catch (...) {
            self->err = std::current_exception();
            *failed = 1;
            return 0.0;
        }
    }
};
struct Context {  // RAII over lbfgs_hip_ctx
    lbfgs_hip_ctx* h = nullptr;
    Context(int device, std::size_t n) {
        const int rc = lbfgs_hip_ctx_create(&h, device, n, nullptr, nullptr, nullptr);
        if (rc != LBFGS_HIP_OK) raise(rc, lbfgs_hip_last_error(nullptr));
        if (lbfgs_hip_abi_version() != LBFGS_HIP_ABI_VERSION) {
            lbfgs_hip_ctx_destroy(h);
            raise(LBFGS_ERR_PARAM, "liblbfgs_hip has another ABI version than this header");
        }
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ~Context() { lbfgs_hip_ctx_destroy(h); }
};
}  // namespace detail

// LbfgsState (lbfgs.rs:425-565): caller-driven stepping; x is refreshed after every propagate (the reference's state holds &mut x)
template <class E>
class LbfgsState {
    std::vector<double>& x_;
    E eval_;
    detail::HostBridge<E> bridge_;
    detail::Context ctx_;
    lbfgs_state* st_ = nullptr;

    [[noreturn]] void fail(int rc) {
        if (rc == LBFGS_ERR_EVALUATE && bridge_.err) std::rethrow_exception(std::exchange(bridge_.err, nullptr));  // the closure's own Err
        detail::raise(rc, st_ ? lbfgs_state_error(st_) : lbfgs_state_error(nullptr));
    }
    void refresh() {
        if (lbfgs_state_download(st_, LBFGS_VEC_X, x_.data()) != LBFGS_OK) fail(LBFGS_HIP_ERR_HIP);
    }

  public:
    LbfgsState(const lbfgs_param& param, int device, std::vector<double>& x, E eval, const lbfgs_evaluator* other = nullptr)
        : x_(x), eval_(std::move(eval)), bridge_{&eval_, nullptr}, ctx_(device, x.size()) {
        lbfgs_evaluator ev{};
        if (other) {
            ev = *other;
        } else {
            ev.kind = LBFGS_EVAL_HOST;
            ev.host = &detail::HostBridge<E>::call;
            ev.user = &bridge_;
        }
        const int rc = lbfgs_build(&st_, ctx_.h, &param, x_.data(), &ev);  // lbfgs.rs:443-481
        if (rc != LBFGS_OK) fail(rc);
    }
    LbfgsState(const LbfgsState&) = delete;
    LbfgsState& operator=(const LbfgsState&) = delete;
    ~LbfgsState() { lbfgs_state_free(st_); }

    bool is_converged() {  // lbfgs.rs:489-494
        int c = 0;
        const int rc = lbfgs_is_converged(st_, &c);
        if (rc != LBFGS_OK) fail(rc);
        return c != 0;
    }
    Progress propagate() {  // lbfgs.rs:503-560
        lbfgs_progress p{};
        const int rc = lbfgs_propagate(st_, &p);
        if (rc != LBFGS_OK) {
            refresh();  // (x names the point the failed search started from, as the reference's copy-based save_state leaves it)
            fail(rc);
        }
        refresh();
        return Progress(st_, x_.size(), p);
    }
    Report report() {  // lbfgs.rs:497-499
        lbfgs_report r{};
        const int rc = lbfgs_get_report(st_, &r);
        if (rc != LBFGS_OK) fail(rc);
        return Report{r.fx, r.xnorm, r.gnorm, static_cast<std::size_t>(r.neval)};
    }
    lbfgs_state* raw() { return st_; }  // for the extension entry points of lbfgs_solver.h
};

class Lbfgs {  // lbfgs.rs:185-421
    lbfgs_param p_;
    int device_ = 0;

  public:
    Lbfgs() { lbfgs_param_default(&p_); }
    const lbfgs_param& param() const { return p_; }

    Lbfgs& with_epsilon(double epsilon) {  // lbfgs.rs:194-199
        detail::require(!std::signbit(epsilon), "Invalid parameter epsilon specified.");
        p_.epsilon = epsilon;
        return *this;
    }
    Lbfgs& with_initial_step_size(double b) {
        detail::require(!std::signbit(b), "Invalid beta parameter for scaling the initial step size.");
        p_.initial_inverse_hessian = b;
        return *this;
    }
    Lbfgs& with_max_step_size(double s) {
        detail::require(!std::signbit(s), "Invalid max_step_size parameter.");
        p_.max_step_size = s;
        return *this;
    }
    Lbfgs& with_damping(bool damped) {
        p_.damping = damped ? 1 : 0;
        return *this;
    }
    // with_orthantwise(c, start, end: impl Into<Option<usize>>) enables OWL-QN (lbfgs.rs:255-274); "only the backtracking line search
    // is available for OWL-QN": LineSearch::find takes it whatever the algorithm says (line.rs:205-211)
    Lbfgs& with_orthantwise(double c, std::size_t start, std::optional<std::size_t> end = std::nullopt) {
        detail::require(!std::signbit(c), "Invalid parameter orthantwise c parameter specified.");
        p_.orthantwise = 1;
        p_.owl_c = c;
        p_.owl_start = start;
        p_.owl_end = end ? static_cast<int64_t>(*end) : -1;
        return *this;
    }
    Lbfgs& with_linesearch_ftol(double ftol) {
        detail::require(ftol >= 0.0, "Invalid parameter ftol specified.");
        p_.ftol = ftol;
        return *this;
    }
    Lbfgs& with_linesearch_gtol(double gtol) {
        detail::require(gtol >= 0.0 && gtol < 1.0 && gtol > p_.ftol, "Invalid parameter gtol specified.");
        p_.gtol = gtol;
        return *this;
    }
    Lbfgs& with_gradient_only() {  // lbfgs.rs:300-306
        p_.gradient_only = 1;
        p_.damping = 1;
        p_.ls_algorithm = LBFGS_LS_BACKTRACKING_STRONGWOLFE;
        return *this;
    }
    Lbfgs& with_max_linesearch(std::size_t n) {
        p_.max_linesearch = n;
        return *this;
    }
    Lbfgs& with_linesearch_xtol(double xtol) {
        detail::require(xtol >= 0.0, "Invalid parameter xtol specified.");
        p_.xtol = xtol;
        return *this;
    }
    Lbfgs& with_linesearch_min_step(double min_step) {
        detail::require(min_step >= 0.0, "Invalid parameter min_step specified.");
        p_.min_step = min_step;
        return *this;
    }
    Lbfgs& with_max_iterations(std::size_t niter) {
        p_.max_iterations = niter;
        return *this;
    }
    Lbfgs& with_max_evaluations(std::size_t neval) {
        p_.max_evaluations = neval;
        return *this;
    }
    Lbfgs& with_fx_delta(double delta, std::size_t past) {  // (sets fields nothing reads: lbfgs.rs:702)
        detail::require(delta >= 0.0, "Invalid parameter delta specified.");
        p_.delta = delta;
        p_.past = past;
        return *this;
    }
    Lbfgs& with_linesearch_algorithm(const std::string& algo) {  // lbfgs.rs:366-383
        if (algo == "MoreThuente") p_.ls_algorithm = LBFGS_LS_MORETHUENTE;
        else if (algo == "BacktrackingArmijo") p_.ls_algorithm = LBFGS_LS_BACKTRACKING_ARMIJO;
        else if (algo == "BacktrackingStrongWolfe") p_.ls_algorithm = LBFGS_LS_BACKTRACKING_STRONGWOLFE;
        else if (algo == "BacktrackingWolfe" || algo == "Backtracking") p_.ls_algorithm = LBFGS_LS_BACKTRACKING_WOLFE;
        else throw Panic(LBFGS_ERR_PARAM, "not implemented: " + algo);  // unimplemented!() lbfgs.rs:379
        return *this;
    }
    // ---- extensions (not reference behaviour)
    Lbfgs& with_m(std::size_t m) {
        detail::require(m >= 1 && m <= 64, "m must be in 1..=64");
        p_.m = m;
        return *this;
    }
    Lbfgs& with_vector_free(bool on = true) {
        p_.vector_free = on ? 1 : 0;
        return *this;
    }
    Lbfgs& with_device(int device) {
        device_ = device;
        return *this;
    }

    // lbfgs.rs:443-481
    template <class E>
    LbfgsState<E> build(std::vector<double>& x, E eval_fn) const {
        return LbfgsState<E>(p_, device_, x, std::move(eval_fn));
    }

    // lbfgs.rs:399-421: `self` is consumed in the reference; here the builder stays usable
    template <class E, class G>
    Report minimize(std::vector<double>& x, E eval_fn, G prgr_fn) const {
        LbfgsState<E> state(p_, device_, x, std::move(eval_fn));
        return run(state, prgr_fn);
    }
    // EXTENSION: one of the device-resident objectives of lbfgs_hip.h (nothing crosses PCIe but the start point and the result)
    template <class G>
    Report minimize_builtin(std::vector<double>& x, const lbfgs_hip_objective& obj, G prgr_fn, int fuse_line_eval = 2) const {
        lbfgs_evaluator ev{};
        ev.kind = LBFGS_EVAL_BUILTIN;
        ev.fuse_line_eval = fuse_line_eval;
        ev.builtin = obj;
        auto unused = [](const double*, double*, std::size_t) { return 0.0; };
        LbfgsState<decltype(unused)> state(p_, device_, x, unused, &ev);
        return run(state, prgr_fn);
    }

  private:
    template <class S, class G>
    static Report run(S& state, G& prgr_fn) {
        for (;;) {  // lbfgs.rs:405-420
            if (state.is_converged()) break;
            const Progress prgr = state.propagate();
            if (prgr_fn(prgr)) break;  // true = cancel (lbfgs.rs:412-416)
        }
        return state.report();
    }
};

inline Lbfgs lbfgs() { return Lbfgs(); }  // lib.rs:74

// default_evaluate (lib.rs:79-94): the N-dimensional Rosenbrock-like test function of the crate's docs
inline auto default_evaluate() {
    return [](const double* x, double* gx, std::size_t n) {
        double fx = 0.0;
        for (std::size_t i = 0; i + 1 < n; i += 2) {
            const double t1 = 1.0 - x[i];
            const double t2 = 10.0 * (x[i + 1] - x[i] * x[i]);
            gx[i + 1] = 20.0 * t2;
            gx[i] = -2.0 * (x[i] * gx[i + 1] + t1);
            fx += t1 * t1 + t2 * t2;
        }
        return fx;
    };
}

// default_progress (lib.rs:101-114): prints what the crate's monitor prints, never cancels
inline auto default_progress() {
    return [](const Progress& prgr) {
        std::printf("Iteration %zu, Evaluation %zu:\n", prgr.niter, prgr.neval);
        std::printf(" fx = %-12.6f xnorm = %-12.6f, gnorm = %-12.6f, ls = %zu, step = %g\n", prgr.fx, prgr.xnorm, prgr.gnorm, prgr.ncall,
                    prgr.step);
        return false;
    };
}

}  // namespace liblbfgs
#endif
