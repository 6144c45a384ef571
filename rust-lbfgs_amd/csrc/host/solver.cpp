// rust-lbfgs_amd/csrc/host/solver.cpp -- host orchestration of L-BFGS / OWL-QN above the C-ABI.
//
// The reference keeps this layer in Rust: Lbfgs::build / LbfgsState::propagate
// (src/lbfgs.rs:443-566), Problem (src/core.rs:10-218) and the two line searches
// (src/line.rs).  Here it is C++ (no Rust toolchain in this image) and it calls
// nothing but the functions of include/lbfgs_hip.h: every n-vector stays in HBM,
// this file only moves O(1) scalars per call and decides which point is evaluated
// next.  The scalar logic is therefore held to the reference operation by
// operation -- it selects the trial steps, so a one-ulp difference here would
// change every later iterate.
//
// Host synchronisations per iteration: one per line-search trial (f and g.d) and ONE
// after the two-loop, which fetches ||s||^2, ys, yy, the norms, ||d||^2 and the next
// line search's g.d together -- nothing inside the two-loop recursion.
#include "../../../include/lbfgs_solver.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// board slots used by this file
enum Slot {
    S_F = 0,        // f(x) of the last evaluate
    S_DG = 1,       // g.d of the last trial
    S_L1 = 2,       // sum c*|x|          (OWL-QN)
    S_PGN2 = 3,     // ||pg||^2           (OWL-QN)
    S_XN2_OWL = 4,  // ||x||^2            (OWL-QN)
    S_DGINIT = 5,
    // one contiguous block read with ONE synchronisation at the end of an iteration:
    S_UPD = 6,      // ||s||^2, ys, yy, ||x||^2, ||g||^2, s.bs, s.(-g)  (history update)
    S_DNORM2 = 13,  // ||d||^2, g.d                                      (two-loop, last step)
    S_DNORM2C = 15, // ||d||^2, pg.d after the orthant projection        (OWL-QN)
    S_END_BLOCK = 17,
    S_NORMS = 17,   // ||x||^2, ||g||^2
    S_FAILED = 19   // closure failure count (world > 1)
};

inline bool sign_positive(double v) { return !std::signbit(v); }  // f64::is_sign_positive

// ------------------------------------------------------------------------------------------------
// More'-Thuente safeguarded step (line.rs:446-606 `mcstep::update_trial_interval`, :620-709)
// ------------------------------------------------------------------------------------------------
struct Endpoint {
    double st, f, dg;  // step, function value, directional derivative
};

// minimiser of the cubic through (u, fu, du) and (v, fv, dv)            line.rs:620-637
inline double cubic_min(double u, double fu, double du, double v, double fv, double dv) {
    const double d = v - u;
    const double theta = (fu - fv) * 3.0 / d + du + dv;
    const double s = std::fmax(std::fmax(std::fabs(theta), std::fabs(du)), std::fabs(dv));
    const double a = theta / s;
    double gamma = s * std::sqrt(a * a - du / s * (dv / s));
    if (v < u) gamma = -gamma;
    const double p = gamma - du + theta;
    const double q = gamma - du + gamma + dv;
    const double r = p / q;
    return u + r * d;
}

// the guarded variant with fall-back to the interval ends                line.rs:652-680
inline double cubic_min_bounded(double u, double fu, double du, double v, double fv, double dv, double lo,
                                double hi) {
    const double d = v - u;
    const double theta = (fu - fv) * 3.0 / d + du + dv;
    const double s = std::fmax(std::fmax(std::fabs(theta), std::fabs(du)), std::fabs(dv));
    const double a = theta / s;
    double gamma = s * std::sqrt(std::fmax(0.0, a * a - du / s * (dv / s)));
    if (u < v) gamma = -gamma;
    const double p = gamma - dv + theta;
    const double q = gamma - dv + gamma + du;
    const double r = p / q;
    if (r < 0.0 && gamma != 0.0) return v - r * d;
    return (v > u) ? hi : lo;
}

inline double quad_min(double u, double fu, double du, double v, double fv) {  // line.rs:692-695
    const double a = v - u;
    return u + du / ((fu - fv) / a + du) / 2.0 * a;
}

inline double secant_min(double u, double du, double v, double dv) {  // line.rs:706-709
    const double a = u - v;
    return v + dv / (dv - du) * a;
}

// Updates the bracket [best, other] and the trial step t.  Returns nullptr, or the bail! message.
const char* safeguarded_step(Endpoint& best, Endpoint& other, double& t, double ft, double dt, double tmin,
                             double tmax, bool& bracketed) {
    const bool opposite = (dt * (best.dg / std::fabs(best.dg))) < 0.0;  // line.rs:461

    if (bracketed) {  // line.rs:470-481
        if (t <= std::fmin(best.st, other.st) || std::fmax(best.st, other.st) <= t)
            return "The line-search step went out of the interval of uncertainty.";
        if (0.0 <= best.dg * (t - best.st))
            return "The current search direction increases the objective function value.";
        if (tmax < tmin)
            return "A logic error occurred; alternatively, the interval of uncertainty became too small.";
    }

    double next;
    bool clamp_to_bracket;
    if (best.f < ft) {  // case 1: higher value -> minimum bracketed                line.rs:484-498
        bracketed = true;
        const double mc = cubic_min(best.st, best.f, best.dg, t, ft, dt);
        const double mq = quad_min(best.st, best.f, best.dg, t, ft);
        next = (std::fabs(mc - best.st) < std::fabs(mq - best.st)) ? mc : mc + 0.5 * (mq - mc);
        clamp_to_bracket = true;
    } else if (opposite) {  // case 2: lower value, derivative changed sign          line.rs:499-513
        bracketed = true;
        const double mc = cubic_min(best.st, best.f, best.dg, t, ft, dt);
        const double mq = secant_min(best.st, best.dg, t, dt);
        next = (std::fabs(mc - t) > std::fabs(mq - t)) ? mc : mq;
        clamp_to_bracket = false;
    } else if (std::fabs(dt) < std::fabs(best.dg)) {  // case 3: derivative shrinks   line.rs:514-538
        const double mc = cubic_min_bounded(best.st, best.f, best.dg, t, ft, dt, tmin, tmax);
        const double mq = secant_min(best.st, best.dg, t, dt);
        if (bracketed)
            next = (std::fabs(t - mc) < std::fabs(t - mq)) ? mc : mq;
        else
            next = (std::fabs(t - mc) > std::fabs(t - mq)) ? mc : mq;
        clamp_to_bracket = true;
    } else {  // case 4: derivative does not shrink                                   line.rs:539-553
        if (bracketed)
            next = cubic_min(t, ft, dt, other.st, other.f, other.dg);
        else
            next = (best.st < t) ? tmax : tmin;
        clamp_to_bracket = false;
    }

    if (best.f < ft) {  // line.rs:563-579
        other = {t, ft, dt};
    } else {
        if (opposite) other = best;
        best = {t, ft, dt};
    }

    if (tmax < next) next = tmax;  // line.rs:582-587
    if (next < tmin) next = tmin;

    if (bracketed && clamp_to_bracket) {  // line.rs:591-600
        const double mq = best.st + 0.66 * (other.st - best.st);
        if (best.st < other.st) {
            if (mq < next) next = mq;
        } else if (next < mq) {
            next = mq;
        }
    }
    t = next;
    return nullptr;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// LbfgsState + Problem, device resident
// ------------------------------------------------------------------------------------------------
struct lbfgs_state {
    lbfgs_hip_ctx* ctx = nullptr;
    lbfgs_param vars{};
    lbfgs_evaluator eval{};
    lbfgs_hip_shard shard{};
    // Problem (core.rs:10-52)
    lbfgs_hip_vec *x = nullptr, *gx = nullptr, *xp = nullptr, *gp = nullptr, *pg = nullptr, *wp = nullptr,
                  *d = nullptr;
    // The next direction is built HERE and exchanged with d only once IterationData::update's checks have passed: the
    // whole tail of an iteration is enqueued before its scalars are read, and after Err("gx not changed") /
    // Err("x not changed") the reference's d is still the old one (lbfgs.rs:646,655 return before :536-540).
    lbfgs_hip_vec* d_next = nullptr;
    double fx = 0.0;
    uint64_t neval = 0;
    uint64_t owl_start = 0, owl_end = 0;
    bool owl_range_known = false;
    double xnorm2 = 0.0, gnorm2 = 0.0;  // squared norms at the current point
    bool norms_valid = false;
    bool evaluated = false;              // core.rs:48
    double *host_x = nullptr, *host_g = nullptr;  // page-locked staging for the host closure (n_local doubles each)
    // LbfgsState (lbfgs.rs:425-439)
    lbfgs_hip_history* hist = nullptr;
    int end = 0;
    double step = 0.0;
    uint64_t k = 0;
    uint64_t ncall = 0;
    double last_gamma = 0.0;
    uint64_t vector_free_fallbacks = 0;  // iterations whose vector-free direction failed its ||d||^2 check and was redone exactly
    double dginit_next = 0.0;  // g.d of the direction just built (fused into the two-loop's last kernel)
    bool dginit_valid = false;
    // deferred trial points (lbfgs_hip_objective_line_probe): trials of the running search leave x/gx unwritten;
    // the accepted point t_eval is materialised by lbfgs_hip_history_update_from_step
    bool defer_trials = false, point_deferred = false;
    double t_eval = 0.0;
    uint64_t trials = 0;  // trial steps evaluated by the running search
    bool orthant_pending = false;  // OWL-QN: wp of the running search is still to be formed, by its first trial's kernel
    // OWL-QN with a built-in objective: every trial of propagate's search also does IterationData::update for its point
    // (lbfgs_hip_objective_owlqn_trial_update) -- a backtracking search accepts the trial it evaluated last (line.rs:747-777)
    bool update_in_trials = false, update_done = false;
    std::string err, ls_err;

    bool owlqn() const { return vars.orthantwise != 0; }
    const lbfgs_hip_vec* grad_for_direction() const { return owlqn() ? pg : gx; }  // core.rs:96-100
};

namespace {

int fail(lbfgs_state* st, int code, const std::string& msg) {
    st->err = msg;
    return code;
}

int backend(lbfgs_state* st, int rc) {
    if (rc != LBFGS_HIP_OK) st->err = std::string("backend: ") + lbfgs_hip_last_error(st->ctx);
    return rc;
}
#define TRY(expr)                                  \
    do {                                           \
        int rc_ = (expr);                          \
        if (rc_ != LBFGS_OK) return rc_;           \
    } while (0)
#define TRYB(st, expr) TRY(backend(st, (expr)))

// orthantwise.rs:59-67 (global indices)
int owl_range(lbfgs_state* st) {
    if (st->owl_range_known) return LBFGS_OK;
    const uint64_t n = st->shard.n_global;
    uint64_t e = (st->vars.owl_end < 0) ? n : (uint64_t)st->vars.owl_end;
    if (e > n) e = n;
    st->owl_start = st->vars.owl_start;
    st->owl_end = e;
    if (!(st->owl_start < st->owl_end)) {
        char b[128];
        snprintf(b, sizeof(b), "invalid start for orthantwise: %llu (end = %llu)", (unsigned long long)st->owl_start,
                 (unsigned long long)st->owl_end);
        return fail(st, LBFGS_PANIC_OWLQN_RANGE, b);
    }
    st->owl_range_known = true;
    return LBFGS_OK;
}

// The user's closure (core.rs:120).  Leaves f in board[S_F]; g in st->gx.
int call_user_evaluate(lbfgs_state* st) {
    const uint64_t nl = st->shard.n_local;
    const bool multi = st->shard.world > 1;
    if (st->eval.kind == LBFGS_EVAL_BUILTIN)
        return backend(st, lbfgs_hip_objective_eval(&st->eval.builtin, st->x, st->gx, S_F));
    int failed = 0;
    double f = 0.0;
    // A rank-local backend failure (staging buffers, transfers, stream) must not leave the peers alone in the
    // collective below: it is folded into `failed`, this rank still takes part, and the stored error is
    // returned afterwards.
    int local_rc = LBFGS_OK;
    std::string local_err;
    auto local = [&](int rc) {
        if (local_rc == LBFGS_OK && backend(st, rc) != LBFGS_OK) {
            local_rc = rc;
            local_err = st->err;
        }
        return local_rc == LBFGS_OK;
    };
    if (st->eval.kind == LBFGS_EVAL_HOST) {
        if (!st->host_x) {  // pinned, so the two transfers per evaluation run at DMA speed
            void *a = nullptr, *b = nullptr;
            if (local(lbfgs_hip_host_buffer_create(st->ctx, nl * sizeof(double), &a))) st->host_x = static_cast<double*>(a);
            if (local(lbfgs_hip_host_buffer_create(st->ctx, nl * sizeof(double), &b))) st->host_g = static_cast<double*>(b);
        }
        if (local(lbfgs_hip_vec_download(st->x, st->host_x, nl))) {
            f = st->eval.host(st->eval.user, st->host_x, st->host_g, nl, &failed);
            if (!failed) local(lbfgs_hip_vec_upload(st->gx, st->host_g, nl));
        }
    } else if (local(lbfgs_hip_sync(st->ctx))) {
        f = st->eval.device(st->eval.user, lbfgs_hip_vec_ptr(st->x), lbfgs_hip_vec_ptr(st->gx), nl,
                            lbfgs_hip_stream(st->ctx), &failed);
    }
    if (local_rc != LBFGS_OK) failed = 1;
    if (multi) {  // partial f per shard; a failure on any rank fails all
        double v[2] = {failed ? 0.0 : f, failed ? 1.0 : 0.0};
        TRYB(st, lbfgs_hip_scalars_write(st->ctx, S_F, 1, &v[0]));
        TRYB(st, lbfgs_hip_scalars_write(st->ctx, S_FAILED, 1, &v[1]));
        TRYB(st, lbfgs_hip_scalars_allreduce(st->ctx, S_F, 1));
        TRYB(st, lbfgs_hip_scalars_allreduce(st->ctx, S_FAILED, 1));
        double nf = 0.0;
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_FAILED, 1, &nf));
        if (local_rc != LBFGS_OK) return fail(st, local_rc, local_err);
        if (nf != 0.0) return fail(st, LBFGS_ERR_EVALUATE, "evaluate failed");
        return LBFGS_OK;
    }
    if (local_rc != LBFGS_OK) return fail(st, local_rc, local_err);
    if (failed) return fail(st, LBFGS_ERR_EVALUATE, "evaluate failed");
    return backend(st, lbfgs_hip_scalars_write(st->ctx, S_F, 1, &f));
}

// Problem::evaluate (core.rs:119-132) at the current x; optionally also g.d.  One host sync.
int evaluate_here(lbfgs_state* st, bool want_dg, double* dg_out) {
    TRY(call_user_evaluate(st));
    if (st->owlqn()) {
        TRY(owl_range(st));  // start_end() asserts on first use, after the closure ran (orthantwise.rs:64)
        TRYB(st, lbfgs_hip_owlqn_post_eval(st->x, st->gx, st->pg, st->vars.owl_c, st->owl_start, st->owl_end, S_L1));
    }
    if (want_dg) TRYB(st, lbfgs_hip_vecdot(st->gx, st->d, S_DG));  // core.rs:114-116: raw gradient
    double b[5] = {0, 0, 0, 0, 0};
    TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_F, 5, b));
    st->fx = b[S_F];
    if (st->owlqn()) {
        st->fx += b[S_L1];  // core.rs:124
        st->gnorm2 = b[S_PGN2];
        st->xnorm2 = b[S_XN2_OWL];
        st->norms_valid = true;
    } else {
        st->norms_valid = false;
    }
    if (dg_out) *dg_out = b[S_DG];
    st->neval += 1;
    st->evaluated = true;
    return LBFGS_OK;
}

// take_line_step + evaluate + dg_unchecked (line.rs:283-288 / :740-753) for one trial step.
// A deferred trial of a DEVICE closure (lbfgs_evaluator.device_probe): f and g.d at xp + t*d, no vector written.
int probe_device(lbfgs_state* st, double t, double* dg_out) {
    const uint64_t nl = st->shard.n_local;
    int failed = 0;
    double f = 0.0, dg = 0.0;
    int local_rc = backend(st, lbfgs_hip_sync(st->ctx));  // (rank-local failures are folded into `failed`, as in call_user_evaluate)
    const std::string local_err = st->err;
    if (local_rc == LBFGS_OK)
        f = st->eval.device_probe(st->eval.user, lbfgs_hip_vec_ptr(st->xp), lbfgs_hip_vec_ptr(st->d), t, nl,
                                  lbfgs_hip_stream(st->ctx), &dg, &failed);
    else
        failed = 1;
    if (st->shard.world > 1) {  // partial f and g.d per shard; a failure on any rank fails all
        double v[2] = {failed ? 0.0 : f, failed ? 0.0 : dg}, nf = failed ? 1.0 : 0.0;
        static_assert(S_DG == S_F + 1, "f and g.d travel in one all-reduce");
        TRYB(st, lbfgs_hip_scalars_write(st->ctx, S_F, 2, v));
        TRYB(st, lbfgs_hip_scalars_write(st->ctx, S_FAILED, 1, &nf));
        TRYB(st, lbfgs_hip_scalars_allreduce(st->ctx, S_F, 2));
        TRYB(st, lbfgs_hip_scalars_allreduce(st->ctx, S_FAILED, 1));
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_FAILED, 1, &nf));
        if (local_rc != LBFGS_OK) return fail(st, local_rc, local_err);
        if (nf != 0.0) return fail(st, LBFGS_ERR_EVALUATE, "evaluate failed");
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_F, 2, v));
        f = v[0];
        dg = v[1];
    } else {
        if (local_rc != LBFGS_OK) return fail(st, local_rc, local_err);
        if (failed) return fail(st, LBFGS_ERR_EVALUATE, "evaluate failed");
    }
    st->fx = f;
    if (dg_out) *dg_out = dg;
    return LBFGS_OK;
}

// The accepted point of a search whose trials were device probes: x = xp + t*d, gx = grad f(x) -- by the closure's own
// `device_accept`, or by take_line_step + the full evaluate (core.rs:155-158, 119-121).  Neither neval nor fx change: the
// reference evaluated this point once, as the search's last trial, and that is the probe already counted.
int materialise_device_point(lbfgs_state* st) {
    const uint64_t nl = st->shard.n_local;
    if (!st->eval.device_accept) {
        TRYB(st, lbfgs_hip_line_step(st->x, st->xp, st->d, st->t_eval, nullptr, 0, 0));
        const double fx = st->fx;
        const int rc = call_user_evaluate(st);  // (collective when world > 1, like every evaluate)
        st->fx = fx;
        return rc;
    }
    int failed = 0;
    int local_rc = backend(st, lbfgs_hip_sync(st->ctx));
    const std::string local_err = st->err;
    if (local_rc == LBFGS_OK)
        (void)st->eval.device_accept(st->eval.user, lbfgs_hip_vec_ptr(st->xp), lbfgs_hip_vec_ptr(st->d), st->t_eval,
                                     lbfgs_hip_vec_ptr(st->x), lbfgs_hip_vec_ptr(st->gx), nl, lbfgs_hip_stream(st->ctx), &failed);
    else
        failed = 1;
    if (st->shard.world > 1) {
        double nf = failed ? 1.0 : 0.0;
        TRYB(st, lbfgs_hip_scalars_write(st->ctx, S_FAILED, 1, &nf));
        TRYB(st, lbfgs_hip_scalars_allreduce(st->ctx, S_FAILED, 1));
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_FAILED, 1, &nf));
        failed = failed || nf != 0.0;
    }
    if (local_rc != LBFGS_OK) return fail(st, local_rc, local_err);
    if (failed) return fail(st, LBFGS_ERR_EVALUATE, "evaluate failed (device_accept)");
    return LBFGS_OK;
}

int trial(lbfgs_state* st, double t, bool want_dg, double* dg_out) {
    st->trials += 1;
    if (st->defer_trials && st->eval.kind == LBFGS_EVAL_DEVICE) {
        TRY(probe_device(st, t, dg_out));
        st->norms_valid = false;
        st->neval += 1;
        st->t_eval = t;
        st->point_deferred = true;
        return LBFGS_OK;
    }
    if (st->defer_trials) {  // f and g.d only; x and gx stay unwritten until the update
        TRYB(st, lbfgs_hip_objective_line_probe(&st->eval.builtin, st->xp, st->d, t, S_F));
        double b[2];
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_F, 2, b));
        st->fx = b[0];
        if (dg_out) *dg_out = b[1];
        st->norms_valid = false;
        st->neval += 1;
        st->t_eval = t;
        st->point_deferred = true;
        return LBFGS_OK;
    }
    if (st->eval.kind == LBFGS_EVAL_BUILTIN && st->eval.fuse_line_eval && !st->owlqn()) {
        TRYB(st, lbfgs_hip_objective_line_eval(&st->eval.builtin, st->x, st->xp, st->d, t, st->gx, S_F));
        double b[2];
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_F, 2, b));
        st->fx = b[0];
        if (dg_out) *dg_out = b[1];
        st->norms_valid = false;
        st->neval += 1;
        return LBFGS_OK;
    }
    if (st->eval.kind == LBFGS_EVAL_BUILTIN && st->eval.fuse_line_eval && st->owlqn()) {
        // OWL-QN trial in one pass: projected line step + evaluate + x1norm + pseudo-gradient (+ g.d); the first trial of a
        // search also forms the orthant of the new point (line.rs:735, core.rs:167-180) -- see search_backtracking
        TRY(owl_range(st));
        if (st->update_in_trials) {
            const int first = st->orthant_pending ? 1 : 0;
            st->orthant_pending = false;
            TRYB(st, lbfgs_hip_objective_owlqn_trial_update(&st->eval.builtin, st->hist, st->end, st->x, st->xp, st->d, t, st->wp, first,
                                                            st->gx, st->gp, st->pg, st->vars.owl_c, st->owl_start, st->owl_end, S_F, S_UPD));
            st->update_done = true;
        } else if (st->orthant_pending) {
            st->orthant_pending = false;
            TRYB(st, lbfgs_hip_objective_owlqn_first_trial(&st->eval.builtin, st->x, st->xp, st->d, t, st->wp, st->gx, st->pg,
                                                           st->vars.owl_c, st->owl_start, st->owl_end, S_F));
        } else
            TRYB(st, lbfgs_hip_objective_owlqn_line_eval(&st->eval.builtin, st->x, st->xp, st->d, t, st->wp, st->gx, st->pg,
                                                         st->vars.owl_c, st->owl_start, st->owl_end, S_F));
        double b[5];
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_F, 5, b));
        st->fx = b[S_F];
        st->fx += b[S_L1];  // core.rs:124
        st->gnorm2 = b[S_PGN2];
        st->xnorm2 = b[S_XN2_OWL];
        st->norms_valid = true;
        if (dg_out) *dg_out = b[S_DG];
        st->neval += 1;
        st->evaluated = true;
        return LBFGS_OK;
    }
    TRYB(st, lbfgs_hip_line_step(st->x, st->xp, st->d, t, st->owlqn() ? st->wp : nullptr, st->owl_start,
                                 st->owl_end));  // core.rs:155-164
    return evaluate_here(st, want_dg, dg_out);
}

int ensure_norms(lbfgs_state* st) {
    if (st->norms_valid) return LBFGS_OK;
    TRYB(st, lbfgs_hip_norms_sq(st->x, st->owlqn() ? st->pg : st->gx, S_NORMS));  // core.rs:183-194
    double b[2];
    TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_NORMS, 2, b));
    st->xnorm2 = b[0];
    st->gnorm2 = b[1];
    st->norms_valid = true;
    return LBFGS_OK;
}

// Progress::new (core.rs:253-268)
int get_progress(lbfgs_state* st, lbfgs_progress* p) {
    TRY(ensure_norms(st));
    p->fx = st->fx;
    p->xnorm = std::sqrt(st->xnorm2);  // math.rs:73-76
    p->gnorm = std::sqrt(st->gnorm2);
    p->neval = st->neval;
    p->ncall = st->ncall;
    p->step = st->step;
    p->niter = st->k;
    return LBFGS_OK;
}

// Problem::revert (core.rs:201-204): fx and pg are NOT restored
int revert(lbfgs_state* st) {
    TRYB(st, lbfgs_hip_veccpy(st->x, st->xp));
    TRYB(st, lbfgs_hip_veccpy(st->gx, st->gp));
    st->norms_valid = false;
    return LBFGS_OK;
}

// ------------------------------------------------------------------------------------------------
// line searches.  Return LBFGS_OK with *ncall, or a hard error; a search failure (the
// reference's bail!) is reported through *bail and handled by the caller (line.rs:213-220).
// ------------------------------------------------------------------------------------------------
int dginit(lbfgs_state* st, double* out) {  // core.rs:78-92 (the gradient of the SAVED point)
    if (st->dginit_valid) {  // already produced by the kernel that finished d
        st->dginit_valid = false;
        *out = st->dginit_next;
        return LBFGS_OK;
    }
    TRYB(st, lbfgs_hip_vecdot(st->owlqn() ? st->pg : st->gp, st->d, S_DGINIT));
    return backend(st, lbfgs_hip_scalars_read(st->ctx, S_DGINIT, 1, out));
}

int search_morethuente(lbfgs_state* st, double& stp, uint64_t* ncall, const char** bail) {  // line.rs:226-399
    const lbfgs_param& pr = st->vars;
    double dg0;
    TRY(dginit(st, &dg0));
    const double f0 = st->fx;
    const double dgtest = pr.ftol * dg0;
    bool bracketed = false, stage1 = true;
    double width = pr.max_step - pr.min_step;
    double prev_width = 2.0 * width;
    Endpoint best{0.0, f0, dg0}, other{0.0, f0, dg0};

    for (uint64_t count = 1; count < pr.max_linesearch; ++count) {
        double stmin, stmax;
        if (bracketed) {
            stmin = (best.st <= other.st) ? best.st : other.st;
            stmax = (best.st >= other.st) ? best.st : other.st;
        } else {
            stmin = best.st;
            stmax = stp + 4.0 * (stp - best.st);
        }
        if (stp < pr.min_step) stp = pr.min_step;
        if (pr.max_step < stp) stp = pr.max_step;

        // unusual termination: fall back to the best step so far (line.rs:277-281; uinfo is always 0)
        const bool tight = bracketed && (stmax - stmin <= pr.xtol * stmax);
        if ((bracketed && (stp <= stmin || stmax <= stp || pr.max_linesearch <= count + 1)) || tight) stp = best.st;

        double dg;
        int rc = trial(st, stp, true, &dg);
        if (rc == LBFGS_ERR_EVALUATE) {
            *bail = "evaluate failed";
            return LBFGS_OK;
        }
        TRY(rc);
        const double f = st->fx;
        const double ftest1 = f0 + stp * dgtest;

        // the reference re-tests with the CURRENT stp (line.rs:292-302)
        if (bracketed && (stp <= stmin || stmax <= stp)) {
            *bail = "A rounding error occurred; alternatively, no line-search step satisfies the sufficient decrease and curvature conditions.";
            return LBFGS_OK;
        }
        if (tight) {
            *bail = "Relative width of the interval of uncertainty is at most xtol.";
            return LBFGS_OK;
        }
        if (stp == pr.max_step && f <= ftest1 && dg <= dgtest) {
            *bail = "The line-search step became larger than LineSearch::max_step.";
            return LBFGS_OK;
        }
        if (stp == pr.min_step && (ftest1 < f || dgtest <= dg)) {
            *bail = "The line-search step became smaller than LineSearch::min_step.";
            return LBFGS_OK;
        }
        if (std::fabs(dg) <= pr.gtol * -dg0) {  // line.rs:315: the curvature test alone ends the search
            *ncall = count;
            return LBFGS_OK;
        }

        if (stage1 && f <= ftest1 && std::fmin(pr.ftol, pr.gtol) * dg0 <= dg) stage1 = false;  // line.rs:324

        const char* msg;
        if (stage1 && ftest1 < f && f <= best.f) {  // modified function (line.rs:333-363)
            Endpoint bm{best.st, best.f - best.st * dgtest, best.dg - dgtest};
            Endpoint om{other.st, other.f - other.st * dgtest, other.dg - dgtest};
            msg = safeguarded_step(bm, om, stp, f - stp * dgtest, dg - dgtest, stmin, stmax, bracketed);
            if (msg) {
                *bail = msg;
                return LBFGS_OK;
            }
            best = {bm.st, bm.f + bm.st * dgtest, bm.dg + dgtest};
            other = {om.st, om.f + om.st * dgtest, om.dg + dgtest};
        } else {
            msg = safeguarded_step(best, other, stp, f, dg, stmin, stmax, bracketed);
            if (msg) {
                *bail = msg;
                return LBFGS_OK;
            }
        }
        if (!bracketed) continue;
        if (0.66 * prev_width <= std::fabs(other.st - best.st)) stp = best.st + 0.5 * (other.st - best.st);
        prev_width = width;
        width = std::fabs(other.st - best.st);
    }
    *ncall = pr.max_linesearch;  // line.rs:398
    return LBFGS_OK;
}

int search_backtracking(lbfgs_state* st, double& stp, uint64_t* ncall, const char** bail) {  // line.rs:716-784
    const lbfgs_param& pr = st->vars;
    double dg0;
    TRY(dginit(st, &dg0));
    const double dec = 0.5, inc = 2.1;
    const double f0 = st->fx;
    const double dgtest = pr.ftol * dg0;
    const bool owl = st->owlqn();
    // line.rs:735, core.rs:167-180: the orthant of the new point.  With a built-in objective and fused trials it is formed BY
    // the first trial's kernel, which streams xp anyway and reads the pseudo-gradient of the start point before it overwrites
    // it (lbfgs_hip_objective_owlqn_first_trial); a search that runs no trial (max_linesearch <= 1) never reads wp.
    st->orthant_pending = owl && st->eval.kind == LBFGS_EVAL_BUILTIN && st->eval.fuse_line_eval;
    if (owl && !st->orthant_pending) TRYB(st, lbfgs_hip_orthant_select(st->wp, st->xp, st->pg));

    const bool armijo_exit = pr.ls_algorithm == LBFGS_LS_BACKTRACKING_ARMIJO || owl;
    const bool want_dg = !armijo_exit || pr.gradient_only;
    for (uint64_t count = 1; count < pr.max_linesearch; ++count) {
        double dg = 0.0;
        int rc = trial(st, stp, want_dg, &dg);
        if (rc == LBFGS_ERR_EVALUATE) {
            *bail = "evaluate failed";
            return LBFGS_OK;
        }
        TRY(rc);
        double width;
        if (st->fx > f0 + stp * dgtest) {
            width = dec;
        } else if (armijo_exit) {
            *ncall = count;
            return LBFGS_OK;
        } else if (dg < pr.gtol * dg0) {
            width = inc;
        } else if (pr.ls_algorithm == LBFGS_LS_BACKTRACKING_WOLFE) {
            *ncall = count;
            return LBFGS_OK;
        } else if (dg > -pr.gtol * dg0) {
            width = dec;
        } else {
            *ncall = count;
            return LBFGS_OK;
        }
        if (pr.gradient_only && std::fabs(dg) <= -pr.gtol * std::fabs(dg0)) {  // line.rs:768-774
            *ncall = count;
            return LBFGS_OK;
        }
        if (stp < pr.min_step) {  // line.rs:776, :166-177
            *bail = "The line-search step became smaller than LineSearch::min_step.";
            return LBFGS_OK;
        }
        if (stp > pr.max_step) {
            *bail = "The line-search step became larger than LineSearch::max_step.";
            return LBFGS_OK;
        }
        stp *= width;
    }
    *ncall = pr.max_linesearch;
    return LBFGS_OK;
}

// LineSearch::find (line.rs:193-223).  Expects xp/gp to hold the base point.  may_defer: the caller will hand
// the accepted step to lbfgs_hip_history_update_from_step (st->point_deferred tells it to).
int line_search_find(lbfgs_state* st, double& step, uint64_t* ncall, bool may_defer = false) {
    const lbfgs_param& pr = st->vars;
    st->trials = 0;
    st->point_deferred = false;
    st->defer_trials = may_defer && !st->owlqn() &&
                       ((st->eval.kind == LBFGS_EVAL_BUILTIN && st->eval.fuse_line_eval >= 2 &&
                         lbfgs_hip_objective_is_elementwise(&st->eval.builtin)) ||
                        (st->eval.kind == LBFGS_EVAL_DEVICE && st->eval.device_probe != nullptr));
    // OWL-QN (always backtracking: line.rs:204-211): the trials do the history update themselves.  Not with Powell damping (its
    // s.bs would be a ninth sum of that kernel), not for the stand-alone line search (no history).
    st->update_done = false;
    static const bool trial_update_off = getenv("LBFGS_OWL_TRIAL_UPDATE") != nullptr && atoi(getenv("LBFGS_OWL_TRIAL_UPDATE")) == 0;  // (A/B, diagnostics)
    st->update_in_trials = may_defer && st->owlqn() && st->eval.kind == LBFGS_EVAL_BUILTIN && st->eval.fuse_line_eval != 0 &&
                           !st->vars.damping && st->hist != nullptr && !trial_update_off;
    struct Reset {
        lbfgs_state* s;
        ~Reset() { s->defer_trials = false; s->update_in_trials = false; }
    } reset{st};
    if (!sign_positive(step)) {
        char b[96];
        snprintf(b, sizeof(b), "A logic error (negative line-search step: %g) occurred.", step);
        return fail(st, LBFGS_ERR_NEGATIVE_STEP, b);
    }
    const char* bail = nullptr;
    st->ls_err.clear();
    *ncall = 0;
    if (pr.ls_algorithm == LBFGS_LS_MORETHUENTE && !st->owlqn()) {
        if (pr.gradient_only)
            return fail(st, LBFGS_ERR_GRADONLY_MT,
                        "Gradient only optimization is incompatible with MoreThuente line search.");
        TRY(search_morethuente(st, step, ncall, &bail));
    } else {
        TRY(search_backtracking(st, step, ncall, &bail));
    }
    if (bail) {  // swallowed: revert and report 0 calls (line.rs:213-220)
        st->ls_err = bail;
        TRY(revert(st));
        st->point_deferred = false;
        *ncall = 0;
    } else if (st->trials == 0) {
        // max_linesearch <= 1: the loops (line.rs:258, :738) run no trial and x stays the saved point.  Here
        // save_state exchanged buffers instead of copying, so x/gx have to be made equal to xp/gp now.
        TRY(revert(st));
    }
    return LBFGS_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

void lbfgs_param_default(lbfgs_param* p) {  // lbfgs.rs:161-176, line.rs:151-162, orthantwise.rs:47-55
    memset(p, 0, sizeof(*p));
    p->m = 6;
    p->epsilon = 1e-5;
    p->delta = 1e-5;
    p->initial_inverse_hessian = 1.0;
    p->max_step_size = 1.0;
    p->constrain_step_size = 1;
    p->ls_algorithm = LBFGS_LS_MORETHUENTE;
    p->ftol = 1e-4;
    p->gtol = 0.9;
    p->xtol = 2.220446049250313e-16;
    p->min_step = 1e-20;
    p->max_step = 1e20;
    p->max_linesearch = 20;
    p->owl_c = 1.0;
    p->owl_end = -1;
}

static std::string g_build_error;

#ifndef LBFGS_SOLVER_BUILD_ID
#define LBFGS_SOLVER_BUILD_ID "unstamped"
#endif
// (readable from the file as well as through the call: rust-lbfgs_amd/_build.py embedded_id)
static const char lbfgs_solver_build_id_marker[] = "LBFGS_SOLVER_BUILD_ID=" LBFGS_SOLVER_BUILD_ID;
const char* lbfgs_solver_build_id(void) { return lbfgs_solver_build_id_marker + sizeof("LBFGS_SOLVER_BUILD_ID=") - 1; }

void lbfgs_state_free(lbfgs_state* st) {
    if (!st) return;
    lbfgs_hip_history_destroy(st->hist);
    lbfgs_hip_host_buffer_destroy(st->ctx, st->host_x);
    lbfgs_hip_host_buffer_destroy(st->ctx, st->host_g);
    lbfgs_hip_vec* vs[] = {st->x, st->gx, st->xp, st->gp, st->pg, st->wp, st->d, st->d_next};
    for (auto* v : vs) lbfgs_hip_vec_free(v);
    delete st;
}

// Problem::new (core.rs:59-75): zeroed vectors (pg/wp only under OWL-QN), x uploaded, nothing evaluated.
static int problem_new(lbfgs_state** out, lbfgs_hip_ctx* ctx, const lbfgs_param* param, const double* x0,
                       const lbfgs_evaluator* eval, bool with_history) {
    if (!out || !ctx || !param || !eval || (with_history && param->m < 1)) return LBFGS_ERR_PARAM;
    *out = nullptr;
    if (lbfgs_hip_abi_version() != LBFGS_HIP_ABI_VERSION) {  // struct layouts of lbfgs_hip.h are shared by value
        char b[128];
        snprintf(b, sizeof(b), "liblbfgs_hip has ABI version %d, this solver was built against %d", lbfgs_hip_abi_version(),
                 LBFGS_HIP_ABI_VERSION);
        g_build_error = b;
        return LBFGS_ERR_PARAM;
    }
    lbfgs_state* st = new (std::nothrow) lbfgs_state();
    if (!st) return LBFGS_HIP_ERR_NOMEM;
    st->ctx = ctx;
    st->vars = *param;
    st->eval = *eval;
    auto bail_out = [&](int rc) {
        g_build_error = st->err;
        lbfgs_state_free(st);
        return rc;
    };
    int rc = backend(st, lbfgs_hip_get_shard(ctx, &st->shard));
    if (rc != LBFGS_OK) return bail_out(rc);
    lbfgs_hip_vec** need[] = {&st->x, &st->gx, &st->xp, &st->gp, &st->d};
    for (auto** v : need)
        if ((rc = backend(st, lbfgs_hip_vec_alloc(ctx, v))) != LBFGS_OK) return bail_out(rc);
    if (st->owlqn()) {
        if ((rc = backend(st, lbfgs_hip_vec_alloc(ctx, &st->pg))) != LBFGS_OK) return bail_out(rc);
        if ((rc = backend(st, lbfgs_hip_vec_alloc(ctx, &st->wp))) != LBFGS_OK) return bail_out(rc);
    }
    // lbfgs.rs:449: m zeroed (s, y) pairs
    if (with_history && (rc = backend(st, lbfgs_hip_history_create(ctx, (int)param->m, &st->hist))) != LBFGS_OK)
        return bail_out(rc);
    if (with_history && (rc = backend(st, lbfgs_hip_vec_alloc(ctx, &st->d_next))) != LBFGS_OK) return bail_out(rc);
    if ((rc = backend(st, lbfgs_hip_vec_upload(st->x, x0, st->shard.n_local))) != LBFGS_OK) return bail_out(rc);
    *out = st;
    return LBFGS_OK;
}

int lbfgs_build(lbfgs_state** out, lbfgs_hip_ctx* ctx, const lbfgs_param* param, const double* x0,
                const lbfgs_evaluator* eval) {
    lbfgs_state* st = nullptr;
    int rc = problem_new(&st, ctx, param, x0, eval, true);
    if (rc != LBFGS_OK) return rc;
    *out = nullptr;
    auto bail_out = [&](int rc_) {
        g_build_error = st->err;
        lbfgs_state_free(st);
        return rc_;
    };
    if ((rc = evaluate_here(st, false, nullptr)) != LBFGS_OK) return bail_out(rc);  // lbfgs.rs:454
    // lbfgs.rs:457: d = -g (or -pg)
    if ((rc = backend(st, lbfgs_hip_vecncpy(st->d, st->grad_for_direction()))) != LBFGS_OK) return bail_out(rc);
    // lbfgs.rs:461: step = (1/||d||) * h0, and ||d|| = ||g|| exactly (d = -g)
    if ((rc = ensure_norms(st)) != LBFGS_OK) return bail_out(rc);
    st->step = (1.0 / std::sqrt(st->gnorm2)) * param->initial_inverse_hessian;
    // g.d with d = -g is -(g.g) exactly, in any fixed summation order (core.rs:78-92)
    st->dginit_next = -st->gnorm2;
    st->dginit_valid = true;
    st->end = 0;
    st->k = 0;
    st->ncall = 0;
    *out = st;
    return LBFGS_OK;
}

// ---- Problem, stand-alone (core.rs:59-217) ----------------------------------------------------------
int lbfgs_problem_new(lbfgs_state** out, lbfgs_hip_ctx* ctx, const lbfgs_param* param, const double* x0,
                      const lbfgs_evaluator* eval) {
    return problem_new(out, ctx, param, x0, eval, false);
}
int lbfgs_problem_evaluate(lbfgs_state* st) {
    if (!st) return LBFGS_ERR_PARAM;
    st->dginit_valid = false;
    return evaluate_here(st, false, nullptr);
}
int lbfgs_problem_update_search_direction(lbfgs_state* st) {
    if (!st) return LBFGS_ERR_PARAM;
    st->dginit_valid = false;
    return backend(st, lbfgs_hip_vecncpy(st->d, st->grad_for_direction()));
}
int lbfgs_problem_dginit(lbfgs_state* st, double* out) {  // on the CURRENT point (gx / pg), as core.rs:78-92
    if (!st || !out) return LBFGS_ERR_PARAM;
    TRYB(st, lbfgs_hip_vecdot(st->owlqn() ? st->pg : st->gx, st->d, S_DGINIT));
    return backend(st, lbfgs_hip_scalars_read(st->ctx, S_DGINIT, 1, out));
}
int lbfgs_problem_dg_unchecked(lbfgs_state* st, double* out) {
    if (!st || !out) return LBFGS_ERR_PARAM;
    TRYB(st, lbfgs_hip_vecdot(st->gx, st->d, S_DG));
    return backend(st, lbfgs_hip_scalars_read(st->ctx, S_DG, 1, out));
}
int lbfgs_problem_save_state(lbfgs_state* st) {  // copies here: x and gx stay valid for the caller
    if (!st) return LBFGS_ERR_PARAM;
    TRYB(st, lbfgs_hip_veccpy(st->xp, st->x));
    return backend(st, lbfgs_hip_veccpy(st->gp, st->gx));
}
int lbfgs_problem_revert(lbfgs_state* st) { return st ? revert(st) : LBFGS_ERR_PARAM; }
int lbfgs_problem_take_line_step(lbfgs_state* st, double step) {
    if (!st) return LBFGS_ERR_PARAM;
    if (st->owlqn()) TRY(owl_range(st));
    st->norms_valid = false;
    return backend(st, lbfgs_hip_line_step(st->x, st->xp, st->d, step, st->owlqn() ? st->wp : nullptr, st->owl_start,
                                           st->owl_end));
}
int lbfgs_problem_update_orthant_new_point(lbfgs_state* st) {
    if (!st || !st->owlqn()) return LBFGS_ERR_PARAM;
    return backend(st, lbfgs_hip_orthant_select(st->wp, st->xp, st->pg));
}
int lbfgs_problem_constrain_search_direction(lbfgs_state* st) {  // no-op without OWL-QN (core.rs:213-217)
    if (!st) return LBFGS_ERR_PARAM;
    if (!st->owlqn()) return LBFGS_OK;
    TRY(owl_range(st));
    st->dginit_valid = false;
    TRYB(st, lbfgs_hip_constrain_direction(st->d, st->pg, st->owl_start, st->owl_end, S_DNORM2C));
    double c2;
    TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_DNORM2C, 1, &c2));
    if (std::sqrt(c2) == 0.0) return fail(st, LBFGS_PANIC_ZERO_DIRECTION, "invalid direction vector after constraints");
    return LBFGS_OK;
}
int lbfgs_problem_norms(lbfgs_state* st, double* xnorm, double* gnorm) {
    if (!st) return LBFGS_ERR_PARAM;
    TRY(ensure_norms(st));
    if (xnorm) *xnorm = std::sqrt(st->xnorm2);
    if (gnorm) *gnorm = std::sqrt(st->gnorm2);
    return LBFGS_OK;
}
int lbfgs_problem_set_linesearch(lbfgs_state* st, const lbfgs_param* ls) {
    if (!st || !ls) return LBFGS_ERR_PARAM;
    st->vars.ls_algorithm = ls->ls_algorithm;
    st->vars.gradient_only = ls->gradient_only;
    st->vars.ftol = ls->ftol; st->vars.gtol = ls->gtol; st->vars.xtol = ls->xtol;
    st->vars.min_step = ls->min_step; st->vars.max_step = ls->max_step;
    st->vars.max_linesearch = ls->max_linesearch;
    return LBFGS_OK;
}
int lbfgs_problem_status(lbfgs_state* st, double* fx, uint64_t* neval, int* evaluated, int* orthantwise) {
    if (!st) return LBFGS_ERR_PARAM;
    if (fx) *fx = st->fx;
    if (neval) *neval = st->neval;
    if (evaluated) *evaluated = st->evaluated ? 1 : 0;
    if (orthantwise) *orthantwise = st->owlqn() ? 1 : 0;
    return LBFGS_OK;
}

int lbfgs_is_converged(lbfgs_state* st, int* converged) {  // lbfgs.rs:489-494, 697-748
    if (!st || !converged) return LBFGS_ERR_PARAM;
    lbfgs_progress p;
    TRY(get_progress(st, &p));
    const lbfgs_param& v = st->vars;
    *converged = 0;
    if (v.max_iterations != 0 && p.niter >= v.max_iterations) *converged = 1;         // :726-735
    else if (v.max_evaluations != 0 && p.neval >= v.max_evaluations) *converged = 1;  // :739-748
    else if (p.gnorm / std::fmax(p.xnorm, 1.0) <= v.epsilon) *converged = 1;          // :715
    return LBFGS_OK;
}

int lbfgs_line_search(lbfgs_state* st, double* step, uint64_t* ncall) {
    if (!st || !step || !ncall) return LBFGS_ERR_PARAM;
    // stand-alone use (line.rs:8-32 doctest): the base point is the current one
    TRYB(st, lbfgs_hip_veccpy(st->xp, st->x));
    TRYB(st, lbfgs_hip_veccpy(st->gp, st->gx));
    st->dginit_valid = false;
    return line_search_find(st, *step, ncall);
}

int lbfgs_propagate(lbfgs_state* st, lbfgs_progress* out) {  // lbfgs.rs:503-560
    if (!st) return LBFGS_ERR_PARAM;
    if (!st->hist) return fail(st, LBFGS_ERR_PARAM, "this state is a stand-alone Problem (no L-BFGS history)");
    st->k += 1;
    if (st->k == 1) return out ? get_progress(st, out) : LBFGS_OK;  // :507-510

    // save_state (core.rs:207-210) without moving bytes: the current point becomes (xp, gp) by
    // exchanging buffers; every trial then writes x = xp + t*d and a fresh gx.
    TRYB(st, lbfgs_hip_vec_swap(st->x, st->xp));
    TRYB(st, lbfgs_hip_vec_swap(st->gx, st->gp));
    st->norms_valid = false;

    uint64_t ncall = 0;
    const double fx0 = st->fx;  // f at the point the search starts from
    {
        const int rc_ls = line_search_find(st, st->step, &ncall, true);  // :517-521
        if (rc_ls != LBFGS_OK) {
            // a hard error (negative step, gradient-only + More-Thuente, backend / communicator failure): the
            // reference's save_state COPIES (core.rs:207-210), so its x still is the point the search started from.
            // Undo the buffer exchange so that VEC_X names that point again (minimize() hands it back to the caller).
            // fx goes back with it: after a failure in the middle of the search st->fx would be the last TRIAL's value,
            // which belongs to neither x nor xp (lbfgs_get_report / lbfgs_problem_status return fx next to x).
            (void)lbfgs_hip_vec_swap(st->x, st->xp);
            (void)lbfgs_hip_vec_swap(st->gx, st->gp);
            st->fx = fx0;
            st->norms_valid = false;
            st->point_deferred = false;
            return rc_ls;
        }
    }
    st->ncall = ncall;
    const double step_ls = st->step;

    // IterationData::update (:525-533, :640-692), then -- without waiting for its scalars --
    // update_search_direction + two-loop + dnorm (:536-543) and the OWL-QN projection (:554): the
    // device forms gamma = ys/yy itself, so the whole tail of the iteration is enqueued at once and
    // its scalars come back in ONE read.  The reference's early exits (:646, :655) need the scalars
    // first, so whenever they can fire by construction (failed line search => x == xp) or a host
    // decision sits in between (Powell damping), the early read is kept.
    const int damping = st->vars.damping;
    const bool early = damping || !st->ls_err.empty();
    bool updated_by_trial = false;
    if (st->point_deferred && st->eval.kind == LBFGS_EVAL_DEVICE) {  // probes of a device closure: form the accepted point now
        st->point_deferred = false;
        const int rc_acc = materialise_device_point(st);
        if (rc_acc != LBFGS_OK) {  // the closure refused the accepted point: as after a hard search error, x names the start point again
            (void)lbfgs_hip_vec_swap(st->x, st->xp);
            (void)lbfgs_hip_vec_swap(st->gx, st->gp);
            st->fx = fx0;
            st->norms_valid = false;
            return rc_acc;
        }
        TRYB(st, lbfgs_hip_history_update(st->hist, st->end, st->x, st->xp, st->gx, st->gp, st->step, damping, S_UPD));
    } else if (st->point_deferred) {  // the trials left x and gx unwritten: take the accepted step in the update's pass
        TRYB(st, lbfgs_hip_history_update_from_step(st->hist, st->end, &st->eval.builtin, st->x, st->xp, st->d, st->t_eval,
                                                    st->gx, st->gp, st->step, damping, S_UPD));
        st->point_deferred = false;
    } else if (st->update_done && st->ls_err.empty()) {
        // the accepted trial -- the last one the search evaluated -- has filled the slot and left ||s||^2, y.s, y.y at S_UPD
        // (a search that failed was reverted: x == xp, and the update below must find exactly that, lbfgs.rs:646)
        updated_by_trial = true;
    } else
        TRYB(st, lbfgs_hip_history_update(st->hist, st->end, st->x, st->xp, st->gx, st->gp, st->step, damping, S_UPD));
    st->update_done = false;
    double u[S_END_BLOCK - S_UPD] = {0};
    auto check_update = [&]() -> int {
        const double snorm = std::sqrt(u[0]), yy = u[2];
        if (!(snorm != 0.0)) {
            char b[96];
            snprintf(b, sizeof(b), "x not changed with step %g", st->step);
            return fail(st, LBFGS_ERR_X_NOT_CHANGED, b);  // :646
        }
        if (!(yy != 0.0)) return fail(st, LBFGS_ERR_GX_NOT_CHANGED, "gx not changed");  // :655
        return LBFGS_OK;
    };
    if (early) {
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_UPD, 6, u));
        TRY(check_update());
        if (damping) {  // :664-689 (sigma2 = 0.6, sigma3 = 3.0)
            const double sigma2 = 0.6, ys = u[1], sbs = u[5];
            if (ys < (1.0 - sigma2) * sbs) {  // case 1: y is replaced; ys and gamma are NOT refreshed (:656, :691)
                const double theta = sigma2 * sbs / (sbs - ys);
                TRYB(st, lbfgs_hip_history_damp(st->hist, st->end, st->gp, st->step, theta));
            }
            // case 2 (:681-685) computes a vector and drops it: nothing to do
        }
    }
    int new_end = st->end;
    bool projected = false;  // constrain_search_direction already applied by the two-loop's last step?
    bool exact = !st->vars.vector_free;
    if (st->vars.vector_free) {
        // EXTENSION: the recursion in coefficient space.  Its arithmetic works on Gram entries -- sums over whole vectors --
        // and loses the digits the recursion needs when its running vector is a small difference of large basis vectors: a
        // run that blows up (y ~ g: q = -g + alpha*y cancels), one that has converged to rounding level, a history with
        // more vectors than the problem has dimensions.  Nothing in the direction itself shows that, so the coefficient
        // kernel reports two figures with it (include/lbfgs_hip.h): the ||d||^2 its Gram arithmetic PREDICTS, next to the
        // ||d||^2 the combine pass sums over the direction itself, and the CANCELLATION of the running vector in the basis
        // -- the factor by which a dot product taken through Gram entries carries more rounding error than the exact
        // recursion's.  When the prediction is off by more than 1e-8, or the cancellation exceeds VF_MAX_CANCELLATION (the
        // exact recursion's 1e-16 * 1e4 = 1e-12 still leaves the 1e-10 this path is held to), THIS iteration's direction is
        // formed again by the exact recursion (lbfgs.rs:569-604) and counted.  The read below waits for the combine kernel
        // the next read would wait for anyway.
        constexpr double VF_MAX_CANCELLATION = 1e4;
        TRYB(st, lbfgs_hip_two_loop_gram(st->hist, st->d_next, st->grad_for_direction(), st->k - 1, st->end, S_UPD + 1,
                                         S_UPD + 2, S_DNORM2, &new_end));
        double chk[4] = {0.0, 0.0, 0.0, 0.0};
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_DNORM2, 4, chk));
        const double dn2 = chk[0], pred = chk[2], cancel = chk[3];
        const bool trusted = std::fabs(pred - dn2) <= 1e-8 * std::fabs(dn2) && cancel <= VF_MAX_CANCELLATION;  // (NaN: not trusted)
        static const bool vf_trace = getenv("LBFGS_VF_TRACE") != nullptr;  // (diagnostics: one line per iteration)
        if (vf_trace)
            fprintf(stderr, "[lbfgs] vector-free k=%llu: ||d||^2 %.17g predicted %.17g (rel %.2e), cancellation %.3g -> %s\n",
                    (unsigned long long)st->k, dn2, pred, std::fabs(pred - dn2) / std::fabs(dn2), cancel, trusted ? "kept" : "EXACT");
        if (!trusted) {
            st->vector_free_fallbacks += 1;
            exact = true;
        }
    }
    if (!exact) {
    } else if (st->owlqn()) {  // :554 folded into the last step (it streams pg anyway); slots S_DNORM2C follow S_DNORM2
        static_assert(S_DNORM2C == S_DNORM2 + 2, "two_loop_owlqn writes 4 adjacent slots");
        TRYB(st, lbfgs_hip_two_loop_owlqn(st->hist, st->d_next, st->pg, st->k - 1, st->end, S_UPD + 1, S_UPD + 2, S_DNORM2,
                                          st->owl_start, st->owl_end, &new_end));
        projected = true;
    } else  // without OWL-QN the update kernel already summed s_new.(-g): start from it
        TRYB(st, lbfgs_hip_two_loop_from(st->hist, st->d_next, st->grad_for_direction(), st->k - 1, st->end, S_UPD + 1,
                                         S_UPD + 2, S_DNORM2, st->owlqn() ? -1 : S_UPD + 6, &new_end));
    if (st->owlqn() && !projected)  // :554, orthantwise.rs:140-161 (after dnorm, as in the reference)
        TRYB(st, lbfgs_hip_constrain_direction(st->d_next, st->pg, st->owl_start, st->owl_end, S_DNORM2C));
    // plain L-BFGS never writes S_DNORM2C..: asking for them would force the copy path of scalars_read every iteration
    if (updated_by_trial) {
        // (the trial's kernel wrote three of the update's seven slots: asking for the others would force the copy path of
        // scalars_read -- a copy kernel and a stream synchronisation -- on every iteration)
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_UPD, 3, u));
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_DNORM2, S_END_BLOCK - S_DNORM2, u + (S_DNORM2 - S_UPD)));
    } else
        TRYB(st, lbfgs_hip_scalars_read(st->ctx, S_UPD, (st->owlqn() ? S_END_BLOCK : S_DNORM2C) - S_UPD, u));
    if (!early) TRY(check_update());
    TRYB(st, lbfgs_hip_vec_swap(st->d, st->d_next));  // the update succeeded: the new direction takes effect (:536-540)
    st->end = new_end;
    if (!st->owlqn()) {
        st->xnorm2 = u[3];
        st->gnorm2 = u[4];
        st->norms_valid = true;
    }
    st->last_gamma = u[1] / u[2];  // :691 ys/yy (the device formed the same quotient from the board)

    const double dnorm = std::sqrt(u[S_DNORM2 - S_UPD]);  // :543
    if (!sign_positive(dnorm)) return fail(st, LBFGS_ERR_INVALID_DNORM, "invalid norm value");  // :544
    st->step = st->vars.constrain_step_size ? std::fmin(st->vars.max_step_size, dnorm) / dnorm : 1.0;  // :547-551
    if (st->owlqn()) {
        if (std::sqrt(u[S_DNORM2C - S_UPD]) == 0.0)
            return fail(st, LBFGS_PANIC_ZERO_DIRECTION, "invalid direction vector after constraints");
        st->dginit_next = u[S_DNORM2C + 1 - S_UPD];  // pg.d of the projected direction (core.rs:90)
    } else {
        st->dginit_next = u[S_DNORM2 + 1 - S_UPD];   // g.d (core.rs:80)
    }
    st->dginit_valid = true;
    if (out) {
        TRY(get_progress(st, out));
        out->step = step_ls;  // :557
    }
    return LBFGS_OK;
}

int lbfgs_get_report(lbfgs_state* st, lbfgs_report* out) {  // lbfgs.rs:497-499, core.rs:288-298
    if (!st || !out) return LBFGS_ERR_PARAM;
    TRY(ensure_norms(st));
    out->fx = st->fx;
    out->xnorm = std::sqrt(st->xnorm2);
    out->gnorm = std::sqrt(st->gnorm2);
    out->neval = st->neval;
    return LBFGS_OK;
}

const char* lbfgs_state_error(const lbfgs_state* st) { return st ? st->err.c_str() : g_build_error.c_str(); }
const char* lbfgs_state_ls_error(const lbfgs_state* st) { return st ? st->ls_err.c_str() : ""; }

static lbfgs_hip_vec* pick(lbfgs_state* st, int which) {
    switch (which) {
        case LBFGS_VEC_X: return st->x;
        case LBFGS_VEC_GX: return st->gx;
        case LBFGS_VEC_XP: return st->xp;
        case LBFGS_VEC_GP: return st->gp;
        case LBFGS_VEC_PG: return st->pg;
        case LBFGS_VEC_WP: return st->wp;
        case LBFGS_VEC_D: return st->d;
        default: break;
    }
    const int m = st->hist ? (int)st->vars.m : 0;
    if (which >= LBFGS_VEC_S0 && which < LBFGS_VEC_S0 + m) return lbfgs_hip_history_s(st->hist, which - LBFGS_VEC_S0);
    if (which >= LBFGS_VEC_Y0 && which < LBFGS_VEC_Y0 + m) return lbfgs_hip_history_y(st->hist, which - LBFGS_VEC_Y0);
    return nullptr;
}

int lbfgs_state_download(lbfgs_state* st, int which, double* host) {
    if (!st || !host) return LBFGS_ERR_PARAM;
    lbfgs_hip_vec* v = pick(st, which);
    if (!v) return fail(st, LBFGS_ERR_PARAM, "no such vector");
    return backend(st, lbfgs_hip_vec_download(v, host, st->shard.n_local));
}

lbfgs_hip_vec* lbfgs_state_vec(lbfgs_state* st, int which) { return st ? pick(st, which) : nullptr; }

void* lbfgs_state_devptr(lbfgs_state* st, int which) {
    lbfgs_hip_vec* v = st ? pick(st, which) : nullptr;
    return v ? lbfgs_hip_vec_ptr(v) : nullptr;
}

int lbfgs_state_vector_free_fallbacks(lbfgs_state* st, uint64_t* count) {
    if (!st || !count) return LBFGS_ERR_PARAM;
    *count = st->vector_free_fallbacks;
    return LBFGS_OK;
}

int lbfgs_state_info(lbfgs_state* st, uint64_t* k, uint64_t* end, double* step, double* gamma) {
    if (!st) return LBFGS_ERR_PARAM;
    if (k) *k = st->k;
    if (end) *end = (uint64_t)st->end;
    if (step) *step = st->step;
    if (gamma) *gamma = st->last_gamma;
    return LBFGS_OK;
}

lbfgs_hip_history* lbfgs_state_history(lbfgs_state* st) { return st ? st->hist : nullptr; }

int lbfgs_minimize(lbfgs_hip_ctx* ctx, const lbfgs_param* param, double* x, const lbfgs_evaluator* eval,
                   lbfgs_progress_cb progress, void* progress_user, lbfgs_report* report, char* errbuf,
                   size_t errbuf_len) {  // lbfgs.rs:399-421
    lbfgs_state* st = nullptr;
    auto note = [&](const char* m) {
        if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", m);
    };
    int rc = lbfgs_build(&st, ctx, param, x, eval);
    if (rc != LBFGS_OK) {
        note(lbfgs_state_error(nullptr));
        return rc;
    }
    for (;;) {
        int conv = 0;
        if ((rc = lbfgs_is_converged(st, &conv)) != LBFGS_OK) break;
        if (conv) break;
        lbfgs_progress p;
        if ((rc = lbfgs_propagate(st, &p)) != LBFGS_OK) break;
        if (progress && progress(progress_user, &p)) break;
    }
    if (rc == LBFGS_OK && report) rc = lbfgs_get_report(st, report);
    // the caller's x always receives the current point (`x: &mut [f64]` is updated in place)
    int rc2 = lbfgs_state_download(st, LBFGS_VEC_X, x);
    if (rc != LBFGS_OK) note(st->err.c_str());
    else if (rc2 != LBFGS_OK) { note(st->err.c_str()); rc = rc2; }
    lbfgs_state_free(st);
    return rc;
}

}  // extern "C"
