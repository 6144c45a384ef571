/*
 * oracle/lbfgs_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See lbfgs_oracle.h for scope, the parity pin and who may load this.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference crate root).  Operation order is the reference's: dot products are
 * strict left-to-right sums of separately rounded products (math.rs:41), axpy
 * is y + (c*x) with two roundings (math.rs:35).  Compile with
 * -ffp-contract=off so the compiler does not fuse them.
 */
#include "lbfgs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* math.rs:31-82                                                              */
/* ------------------------------------------------------------------------- */

static int g_dot_mode = 0;

void oracle_set_dot_mode(int mode) { g_dot_mode = mode; }

static double dot_pairwise(const double* x, const double* y, size_t n) {
    if (n <= 32) {
        double acc = 0.0;
        for (size_t i = 0; i < n; ++i) acc += x[i] * y[i];
        return acc;
    }
    size_t h = n / 2;
    return dot_pairwise(x, y, h) + dot_pairwise(x + h, y + h, n - h);
}

/* DIAGNOSTIC ONLY (dot mode 1): pairwise sum of a term array; used by every sum of the restatement that the
 * reference forms sequentially -- vecdot, x1norm and the objectives' f -- so that mode 1 perturbs ALL of them. */
static double sum_pairwise(const double* t, size_t n) {
    if (n <= 32) {
        double acc = 0.0;
        for (size_t i = 0; i < n; ++i) acc += t[i];
        return acc;
    }
    size_t h = n / 2;
    return sum_pairwise(t, h) + sum_pairwise(t + h, n - h);
}
/* DIAGNOSTIC ONLY (dot modes 3 and 4): K running sums, term i into sum i % K, joined by a pairwise tree -- the shape a
 * vector unit gives a sum (K = 4) and the shape of a wavefront's lanes (K = 64).  x == NULL: the terms are y[i] alone. */
static double sum_strided(const double* x, const double* y, size_t n, int K) {
    double acc[64];
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    for (size_t i = 0; i < n; ++i) acc[i % (size_t)K] += x ? x[i] * y[i] : y[i];
    for (int w = 1; w < K; w *= 2)
        for (int k = 0; k + w < K; k += 2 * w) acc[k] += acc[k + w];
    return acc[0];
}
int oracle_dot_mode(void) { return g_dot_mode; }
double oracle_sum_terms(const double* t, size_t n) {
    if (g_dot_mode == 1) return sum_pairwise(t, n);
    if (g_dot_mode == 3) return sum_strided(NULL, t, n, 4);
    if (g_dot_mode == 4) return sum_strided(NULL, t, n, 64);
    if (g_dot_mode == 2) { /* diagnostic: the reference's running sum, from the last term down */
        double acc = 0.0;
        for (size_t i = n; i-- > 0;) acc += t[i];
        return acc;
    }
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) acc += t[i];
    return acc;
}

/* math.rs:33-37  y += c*x */
void oracle_vecadd(double* y, const double* x, double c, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] += c * x[i];
}

/* math.rs:40-42  s = sum_i x_i*y_i, sequential */
double oracle_vecdot(const double* x, const double* y, size_t n) {
    if (g_dot_mode == 1) return dot_pairwise(x, y, n);
    if (g_dot_mode == 3) return sum_strided(x, y, n, 4);
    if (g_dot_mode == 4) return sum_strided(x, y, n, 64);
    if (g_dot_mode == 2) { /* diagnostic: the same terms, summed from the last one down */
        double acc = 0.0;
        for (size_t i = n; i-- > 0;) acc += x[i] * y[i];
        return acc;
    }
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) acc += x[i] * y[i];
    return acc;
}

/* math.rs:45-49 */
void oracle_vecscale(double* y, double c, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] *= c;
}

/* math.rs:52-56 */
void oracle_veccpy(double* y, const double* x, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = x[i];
}

/* math.rs:59-63 */
void oracle_vecncpy(double* y, const double* x, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = -x[i];
}

/* math.rs:66-70 */
void oracle_vecdiff(double* z, const double* x, const double* y, size_t n) {
    for (size_t i = 0; i < n; ++i) z[i] = x[i] - y[i];
}

/* math.rs:73-76 */
double oracle_vec2norm(const double* x, size_t n) { return sqrt(oracle_vecdot(x, x, n)); }

/* math.rs:79-81 */
double oracle_vec2norminv(const double* x, size_t n) { return 1.0 / oracle_vec2norm(x, n); }

/* ------------------------------------------------------------------------- */
/* orthantwise.rs                                                             */
/* ------------------------------------------------------------------------- */

/* orthantwise.rs:174-180: NaN and +-0 -> 0, else +-1 */
double oracle_signum(double x) {
    if (isnan(x) || x == 0.0) return 0.0;
    return signbit(x) ? -1.0 : 1.0;
}

/* orthantwise.rs:59-67 */
int oracle_owl_range(const oracle_param* p, size_t n, size_t* start, size_t* end) {
    size_t s = (size_t)p->owl_start;
    size_t e = (p->owl_end < 0) ? n : (size_t)p->owl_end;
    if (e > n) e = n;
    *start = s;
    *end = e;
    return (s < e) ? ORACLE_OK : ORACLE_PANIC_OWLQN_RANGE;
}

/* orthantwise.rs:70-79: the multiply by c is inside the sum */
double oracle_x1norm(double c, size_t start, size_t end, const double* x) {
    if (g_dot_mode != 0 && end > start) {  /* diagnostic: the same terms, pairwise / in reverse */
        double* t = (double*)malloc((end - start) * sizeof(double));
        if (t) {
            for (size_t i = start; i < end; ++i) t[i - start] = c * fabs(x[i]);
            double r = oracle_sum_terms(t, end - start);
            free(t);
            return r;
        }
    }
    double s = 0.0;
    for (size_t i = start; i < end; ++i) s += c * fabs(x[i]);
    return s;
}

/* orthantwise.rs:82-112 */
void oracle_pseudo_gradient(double c, size_t start, size_t end, double* pg, const double* x,
                            const double* g, size_t n) {
    for (size_t i = 0; i < start; ++i) pg[i] = g[i];
    for (size_t i = start; i < end; ++i) {
        if (x[i] != 0.0) {
            /* f64::signum: +-1 by sign bit (x is non-zero, NaN propagates) */
            double sg = isnan(x[i]) ? x[i] : (signbit(x[i]) ? -1.0 : 1.0);
            pg[i] = g[i] + sg * c;
        } else {
            double right_partial = g[i] + c;
            double left_partial = g[i] - c;
            if (right_partial < 0.0)
                pg[i] = right_partial;
            else if (left_partial > 0.0)
                pg[i] = left_partial;
            else
                pg[i] = 0.0;
        }
    }
    for (size_t i = end; i < n; ++i) pg[i] = g[i];
}

/* orthantwise.rs:165-171 `project` on [start,end); negate_w selects the
 * `-pg` orthant of constrain_search_direction (orthantwise.rs:145). */
void oracle_project(double* x, const double* w, size_t start, size_t end, int negate_w) {
    for (size_t i = start; i < end; ++i) {
        double wi = negate_w ? -w[i] : w[i];
        if (oracle_signum(x[i]) != oracle_signum(wi)) x[i] = 0.0;
    }
}

/* core.rs:167-180: over ALL i, not just [start,end) */
void oracle_orthant_select(double* wp, const double* xp, const double* pg, size_t n) {
    for (size_t i = 0; i < n; ++i)
        wp[i] = (xp[i] == 0.0) ? oracle_signum(-pg[i]) : oracle_signum(xp[i]);
}

/* ------------------------------------------------------------------------- */
/* lbfgs.rs:569-604 two-loop recursion                                         */
/* ------------------------------------------------------------------------- */
size_t oracle_two_loop(double* const* s, double* const* y, const double* ys, double* alpha,
                       double* d, double gamma, size_t m, size_t k, size_t end, size_t n) {
    end = (end + 1) % m;
    size_t j = end;
    size_t bound = (m < k) ? m : k;

    for (size_t it = 0; it < bound; ++it) {
        j = (j + m - 1) % m;
        alpha[j] = oracle_vecdot(s[j], d, n) / ys[j];   /* lbfgs.rs:587 */
        oracle_vecadd(d, y[j], -alpha[j], n);           /* lbfgs.rs:589 */
    }
    oracle_vecscale(d, gamma, n);                       /* lbfgs.rs:591 */

    for (size_t it = 0; it < bound; ++it) {
        double beta = oracle_vecdot(y[j], d, n) / ys[j];/* lbfgs.rs:597 */
        oracle_vecadd(d, s[j], alpha[j] - beta, n);     /* lbfgs.rs:599 */
        j = (j + 1) % m;
    }
    return end;
}

/* ------------------------------------------------------------------------- */
/* lbfgs.rs:640-692 IterationData::update                                      */
/* ------------------------------------------------------------------------- */
int oracle_history_update(double* s, double* y, double* ys_out, double* gamma, const double* x,
                          const double* xp, const double* gx, const double* gp, double step,
                          int damping, size_t n, double* aux) {
    oracle_vecdiff(s, x, xp, n);                        /* :644 */
    double dn = oracle_vec2norm(s, n);                  /* :645 */
    if (aux) { aux[0] = dn; aux[1] = aux[2] = aux[3] = aux[4] = 0.0; }
    if (!(dn != 0.0)) return ORACLE_ERR_X_NOT_CHANGED;  /* :646 ensure!(d != 0.0) */
    oracle_vecdiff(y, gx, gp, n);                       /* :647 */

    double ys = oracle_vecdot(y, s, n);                 /* :653 */
    double yy = oracle_vecdot(y, y, n);                 /* :654 */
    if (aux) { aux[1] = ys; aux[2] = yy; }
    if (!(yy != 0.0)) return ORACLE_ERR_GX_NOT_CHANGED; /* :655 */
    *ys_out = ys;                                       /* :656 stored BEFORE damping */

    const double sigma2 = 0.6, sigma3 = 3.0;            /* :664-665 */
    if (damping) {
        double* bs = (double*)malloc(n * sizeof(double));
        if (!bs) return ORACLE_ERR_ALLOC;
        oracle_veccpy(bs, gp, n);                       /* :670 gp.to_vec() */
        oracle_vecscale(bs, -step, n);                  /* :671 */
        double sbs = oracle_vecdot(s, bs, n);           /* :673 */
        if (aux) aux[3] = sbs;
        if (ys < (1.0 - sigma2) * sbs) {                /* :675 case 1 */
            double theta = sigma2 * sbs / (sbs - ys);
            oracle_vecscale(bs, 1.0 - theta, n);
            oracle_vecadd(bs, y, theta, n);
            oracle_veccpy(y, bs, n);                    /* :680 y replaced */
            if (aux) aux[4] = 1.0;
        } else if (ys > (1.0 + sigma3) * sbs) {         /* :681 case 2: result discarded */
            double theta = sigma3 * sbs / (ys - sbs);
            oracle_vecscale(bs, 1.0 - theta, n);
            oracle_vecadd(bs, y, theta, n);
            if (aux) aux[4] = 2.0;
        }
        free(bs);
    }
    *gamma = ys / yy;                                   /* :691 */
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* defaults: lbfgs.rs:161-176, line.rs:151-162, orthantwise.rs:47-55           */
/* ------------------------------------------------------------------------- */
void oracle_param_default(oracle_param* p) {
    memset(p, 0, sizeof(*p));
    p->m = 6;
    p->epsilon = 1e-5;
    p->past = 0;
    p->delta = 1e-5;
    p->max_iterations = 0;
    p->max_evaluations = 0;
    p->initial_inverse_hessian = 1.0;
    p->max_step_size = 1.0;
    p->damping = 0;
    p->constrain_step_size = 1;
    p->ls_algorithm = ORACLE_LS_MORETHUENTE;
    p->gradient_only = 0;
    p->ftol = 1e-4;
    p->gtol = 0.9;
    p->xtol = 2.220446049250313e-16; /* f64::EPSILON */
    p->min_step = 1e-20;
    p->max_step = 1e20;
    p->max_linesearch = 20;
    p->orthantwise = 0;
    p->owl_c = 1.0;
    p->owl_start = 0;
    p->owl_end = -1;
}

/* ------------------------------------------------------------------------- */
/* Problem (core.rs:10-218) + LbfgsState (lbfgs.rs:425-439)                    */
/* ------------------------------------------------------------------------- */
struct oracle_state {
    oracle_param vars;
    /* Problem */
    size_t n;
    double* x;    /* borrowed from the caller (core.rs:16) */
    double fx;
    double *gx, *xp, *gp, *pg, *wp, *d;
    oracle_eval_cb eval;
    void* user;
    uint64_t neval;
    size_t owl_start, owl_end;
    /* LbfgsState */
    size_t end;
    double step;
    uint64_t k;
    uint64_t ncall;
    double **s, **y;
    double *ys, *alpha;
    double last_gamma;
    int ls_failed;
    char ls_error[256];
};

static int sign_positive(double v) { return !signbit(v); } /* f64::is_sign_positive */

/* core.rs:119-132 */
static int problem_evaluate(oracle_state* st) {
    int failed = 0;
    double f = st->eval(st->user, st->x, st->gx, st->n, &failed);
    if (failed) return ORACLE_ERR_EVALUATE;
    st->fx = f;
    if (st->vars.orthantwise) {
        /* orthantwise.rs:64: start_end() asserts on first use, i.e. after the closure ran */
        int rc = oracle_owl_range(&st->vars, st->n, &st->owl_start, &st->owl_end);
        if (rc != ORACLE_OK) return rc;
        st->fx += oracle_x1norm(st->vars.owl_c, st->owl_start, st->owl_end, st->x);
        oracle_pseudo_gradient(st->vars.owl_c, st->owl_start, st->owl_end, st->pg, st->x, st->gx,
                               st->n);
    }
    st->neval += 1;
    return ORACLE_OK;
}

/* core.rs:78-92 (the warn! on dginit > 0 has no effect on the arithmetic) */
static double problem_dginit(const oracle_state* st) {
    return st->vars.orthantwise ? oracle_vecdot(st->pg, st->d, st->n)
                                : oracle_vecdot(st->gx, st->d, st->n);
}

/* core.rs:114-116: always the RAW gradient, also under OWL-QN */
static double problem_dg_unchecked(const oracle_state* st) {
    return oracle_vecdot(st->gx, st->d, st->n);
}

/* core.rs:95-101 */
static void problem_update_search_direction(oracle_state* st) {
    oracle_vecncpy(st->d, st->vars.orthantwise ? st->pg : st->gx, st->n);
}

/* core.rs:155-164 */
static void problem_take_line_step(oracle_state* st, double step) {
    oracle_veccpy(st->x, st->xp, st->n);
    oracle_vecadd(st->x, st->d, step, st->n);
    if (st->vars.orthantwise)
        oracle_project(st->x, st->wp, st->owl_start, st->owl_end, 0); /* orthantwise.rs:118-133 */
}

/* core.rs:183-194 */
static double problem_gnorm(const oracle_state* st) {
    return oracle_vec2norm(st->vars.orthantwise ? st->pg : st->gx, st->n);
}
static double problem_xnorm(const oracle_state* st) { return oracle_vec2norm(st->x, st->n); }

/* core.rs:201-204: fx and pg are NOT restored */
static void problem_revert(oracle_state* st) {
    oracle_veccpy(st->x, st->xp, st->n);
    oracle_veccpy(st->gx, st->gp, st->n);
}

/* core.rs:207-210 */
static void problem_save_state(oracle_state* st) {
    oracle_veccpy(st->xp, st->x, st->n);
    oracle_veccpy(st->gp, st->gx, st->n);
}

/* ------------------------------------------------------------------------- */
/* line.rs:620-709 interpolation helpers                                       */
/* ------------------------------------------------------------------------- */

/* f64::max / f64::min: ignore a NaN operand */
static double rmax(double a, double b) { return fmax(a, b); }
static double rmin(double a, double b) { return fmin(a, b); }

/* line.rs:620-637 */
static double cubic_minimizer(double u, double fu, double du, double v, double fv, double dv) {
    double d = v - u;
    double theta = (fu - fv) * 3.0 / d + du + dv;
    double p = fabs(theta);
    double q = fabs(du);
    double r = fabs(dv);
    double s = rmax(rmax(p, q), r);
    double a = theta / s;
    double gamma = s * sqrt(a * a - du / s * (dv / s)); /* no max(0,.) guard here (:629) */
    if (v < u) gamma = -gamma;
    p = gamma - du + theta;
    q = gamma - du + gamma + dv;
    r = p / q;
    return u + r * d;
}

/* line.rs:652-680 */
static double cubic_minimizer2(double u, double fu, double du, double v, double fv, double dv,
                               double xmin, double xmax) {
    double d = v - u;
    double theta = (fu - fv) * 3.0 / d + du + dv;
    double p = fabs(theta);
    double q = fabs(du);
    double r = fabs(dv);
    double s = rmax(rmax(p, q), r);
    double a = theta / s;
    double gamma = s * sqrt(rmax(0.0, a * a - du / s * (dv / s)));
    if (u < v) gamma = -gamma;
    p = gamma - dv + theta;
    q = gamma - dv + gamma + du;
    r = p / q;
    if (r < 0.0 && gamma != 0.0) return v - r * d;
    if (v > u) return xmax;
    return xmin;
}

/* line.rs:692-695 */
static double quard_minimizer(double u, double fu, double du, double v, double fv) {
    double a = v - u;
    return u + du / ((fu - fv) / a + du) / 2.0 * a;
}

/* line.rs:706-709 */
static double quard_minimizer2(double u, double du, double v, double dv) {
    double a = u - v;
    return v + dv / (dv - du) * a;
}

/* line.rs:446-606 mcstep::update_trial_interval.  Returns 0, or 1 with msg on bail!. */
static int update_trial_interval(double* x, double* fx, double* dx, double* y, double* fy,
                                 double* dy, double* t, double ft, double dt, double tmin,
                                 double tmax, int* brackt, const char** msg) {
    int dsign = (dt * (*dx / fabs(*dx))) < 0.0; /* :461 */
    double mc, mq, newt;
    int bound;

    if (*brackt) { /* :470-481 */
        if (*t <= rmin(*x, *y) || rmax(*x, *y) <= *t) {
            *msg = "The line-search step went out of the interval of uncertainty.";
            return 1;
        } else if (0.0 <= *dx * (*t - *x)) {
            *msg = "The current search direction increases the objective function value.";
            return 1;
        } else if (tmax < tmin) {
            *msg = "A logic error occurred; alternatively, the interval of uncertainty became too small.";
            return 1;
        }
    }

    if (*fx < ft) { /* case 1 :484-498 */
        *brackt = 1;
        mc = cubic_minimizer(*x, *fx, *dx, *t, ft, dt);
        mq = quard_minimizer(*x, *fx, *dx, *t, ft);
        if (fabs(mc - *x) < fabs(mq - *x))
            newt = mc;
        else
            newt = mc + 0.5 * (mq - mc);
        bound = 1;
    } else if (dsign) { /* case 2 :499-513 */
        *brackt = 1;
        mc = cubic_minimizer(*x, *fx, *dx, *t, ft, dt);
        mq = quard_minimizer2(*x, *dx, *t, dt);
        if (fabs(mc - *t) > fabs(mq - *t))
            newt = mc;
        else
            newt = mq;
        bound = 0;
    } else if (fabs(dt) < fabs(*dx)) { /* case 3 :514-538 */
        mc = cubic_minimizer2(*x, *fx, *dx, *t, ft, dt, tmin, tmax);
        mq = quard_minimizer2(*x, *dx, *t, dt);
        if (*brackt) {
            if (fabs(*t - mc) < fabs(*t - mq))
                newt = mc;
            else
                newt = mq;
        } else if (fabs(*t - mc) > fabs(*t - mq)) {
            newt = mc;
        } else {
            newt = mq;
        }
        bound = 1;
    } else { /* case 4 :539-553 */
        if (*brackt)
            newt = cubic_minimizer(*t, ft, dt, *y, *fy, *dy);
        else if (*x < *t)
            newt = tmax;
        else
            newt = tmin;
        bound = 0;
    }

    if (*fx < ft) { /* :563-579 */
        *y = *t;
        *fy = ft;
        *dy = dt;
    } else {
        if (dsign) {
            *y = *x;
            *fy = *fx;
            *dy = *dx;
        }
        *x = *t;
        *fx = ft;
        *dx = dt;
    }

    if (tmax < newt) newt = tmax; /* :582-587 */
    if (newt < tmin) newt = tmin;

    if (*brackt && bound) { /* :591-600 */
        mq = *x + 0.66 * (*y - *x);
        if (*x < *y) {
            if (mq < newt) newt = mq;
        } else if (newt < mq) {
            newt = mq;
        }
    }
    *t = newt;
    return 0;
}

/* line.rs:226-399.  Returns >=0 ncall on Ok, -1 on bail! (msg set), or
 * ORACLE_ERR_EVALUATE-coded failure via *eval_failed. */
static long long line_search_morethuente(oracle_state* st, double* stp, const char** msg,
                                         int* eval_failed) {
    const oracle_param* pr = &st->vars;
    double dginit = problem_dginit(st);
    int brackt = 0, stage1 = 1, uinfo = 0;
    double finit = st->fx;
    double dgtest = pr->ftol * dginit;
    double width = pr->max_step - pr->min_step;
    double prev_width = 2.0 * width;
    double stx = 0.0, sty = 0.0;
    double fx = finit, fy = finit;
    double dgy = dginit, dgx = dginit;

    for (uint64_t count = 1; count < pr->max_linesearch; ++count) { /* :258 */
        double stmin, stmax;
        if (brackt) {
            stmin = (stx <= sty) ? stx : sty;
            stmax = (stx >= sty) ? stx : sty;
        } else {
            stmin = stx;
            stmax = *stp + 4.0 * (*stp - stx);
        }
        if (*stp < pr->min_step) *stp = pr->min_step; /* :268-273 */
        if (pr->max_step < *stp) *stp = pr->max_step;

        if ((brackt && (*stp <= stmin || stmax <= *stp || pr->max_linesearch <= count + 1 || uinfo != 0)) ||
            (brackt && stmax - stmin <= pr->xtol * stmax)) { /* :277-281 */
            *stp = stx;
        }

        problem_take_line_step(st, *stp); /* :283 */
        if (problem_evaluate(st) != ORACLE_OK) { /* :286 */
            *eval_failed = 1;
            *msg = "evaluate failed";
            return -1;
        }
        double f = st->fx;
        double dg = problem_dg_unchecked(st);
        double ftest1 = finit + *stp * dgtest;

        if (brackt && (*stp <= stmin || stmax <= *stp || uinfo != 0)) { /* :292 */
            *msg = "A rounding error occurred; alternatively, no line-search step satisfies the sufficient decrease and curvature conditions.";
            return -1;
        }
        if (brackt && stmax - stmin <= pr->xtol * stmax) { /* :300 */
            *msg = "Relative width of the interval of uncertainty is at most xtol.";
            return -1;
        }
        if (*stp == pr->max_step && f <= ftest1 && dg <= dgtest) { /* :305 */
            *msg = "The line-search step became larger than LineSearch::max_step.";
            return -1;
        }
        if (*stp == pr->min_step && (ftest1 < f || dgtest <= dg)) { /* :310 */
            *msg = "The line-search step became smaller than LineSearch::min_step.";
            return -1;
        }

        if (fabs(dg) <= pr->gtol * -dginit) { /* :315 curvature only */
            return (long long)count;
        } else if (f <= ftest1 && fabs(dg) <= pr->gtol * -dginit) { /* :318 unreachable */
            return (long long)count;
        } else {
            if (stage1 && f <= ftest1 && rmin(pr->ftol, pr->gtol) * dginit <= dg) stage1 = 0; /* :324 */

            if (stage1 && ftest1 < f && f <= fx) { /* :333 */
                double fm = f - *stp * dgtest;
                double fxm = fx - stx * dgtest;
                double fym = fy - sty * dgtest;
                double dgm = dg - dgtest;
                double dgxm = dgx - dgtest;
                double dgym = dgy - dgtest;
                if (update_trial_interval(&stx, &fxm, &dgxm, &sty, &fym, &dgym, stp, fm, dgm, stmin,
                                          stmax, &brackt, msg))
                    return -1;
                uinfo = 0;
                fx = fxm + stx * dgtest; /* :360-363 */
                fy = fym + sty * dgtest;
                dgx = dgxm + dgtest;
                dgy = dgym + dgtest;
            } else {
                if (update_trial_interval(&stx, &fx, &dgx, &sty, &fy, &dgy, stp, f, dg, stmin, stmax,
                                          &brackt, msg))
                    return -1;
                uinfo = 0;
            }

            if (!brackt) continue; /* :382-384 */
            if (0.66 * prev_width <= fabs(sty - stx)) *stp = stx + 0.5 * (sty - stx); /* :386 */
            prev_width = width;
            width = fabs(sty - stx);
        }
    }
    return (long long)pr->max_linesearch; /* :398 */
}

/* line.rs:716-784 */
static long long line_search_backtracking(oracle_state* st, double* stp, const char** msg,
                                          int* eval_failed) {
    const oracle_param* pr = &st->vars;
    double dginit = problem_dginit(st);
    const double dec = 0.5, inc = 2.1;
    double finit = st->fx;
    double dgtest = pr->ftol * dginit;
    int orthantwise = pr->orthantwise;
    if (orthantwise) oracle_orthant_select(st->wp, st->xp, st->pg, st->n); /* :735 */

    double width;
    for (uint64_t count = 1; count < pr->max_linesearch; ++count) {
        problem_take_line_step(st, *stp);
        if (problem_evaluate(st) != ORACLE_OK) {
            *eval_failed = 1;
            *msg = "evaluate failed";
            return -1;
        }
        if (st->fx > finit + *stp * dgtest) { /* :745 */
            width = dec;
        } else if (pr->ls_algorithm == ORACLE_LS_BACKTRACKING_ARMIJO || orthantwise) {
            return (long long)count; /* :750 */
        } else {
            double dg = problem_dg_unchecked(st);
            if (dg < pr->gtol * dginit) {
                width = inc;
            } else if (pr->ls_algorithm == ORACLE_LS_BACKTRACKING_WOLFE) {
                return (long long)count;
            } else if (dg > -pr->gtol * dginit) {
                width = dec;
            } else {
                return (long long)count;
            }
        }
        if (pr->gradient_only) { /* :768-774 */
            double dg = problem_dg_unchecked(st);
            if (fabs(dg) <= -pr->gtol * fabs(dginit)) return (long long)count;
        }
        if (*stp < pr->min_step) { /* :776 validate_step, line.rs:166-177 */
            *msg = "The line-search step became smaller than LineSearch::min_step.";
            return -1;
        }
        if (*stp > pr->max_step) {
            *msg = "The line-search step became larger than LineSearch::max_step.";
            return -1;
        }
        *stp *= width; /* :777 */
    }
    return (long long)pr->max_linesearch;
}

/* line.rs:193-223 */
int oracle_line_search(oracle_state* st, double* step, uint64_t* ncall) {
    const oracle_param* pr = &st->vars;
    if (!sign_positive(*step)) return ORACLE_ERR_NEGATIVE_STEP; /* :198 */
    const char* msg = "";
    int eval_failed = 0;
    long long ls;
    st->ls_failed = 0;
    st->ls_error[0] = 0;
    if (pr->ls_algorithm == ORACLE_LS_MORETHUENTE && !pr->orthantwise) {
        if (pr->gradient_only) return ORACLE_ERR_GRADONLY_MT; /* :208 */
        ls = line_search_morethuente(st, step, &msg, &eval_failed);
    } else {
        ls = line_search_backtracking(st, step, &msg, &eval_failed);
    }
    if (ls < 0) { /* :213-220 swallow, revert, Ok(0) */
        problem_revert(st);
        st->ls_failed = 1;
        snprintf(st->ls_error, sizeof(st->ls_error), "%s", msg);
        ls = 0;
    }
    *ncall = (uint64_t)ls;
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* lbfgs.rs:443-481 build                                                      */
/* ------------------------------------------------------------------------- */
static double* zalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

void oracle_free(oracle_state* st) {
    if (!st) return;
    free(st->gx); free(st->xp); free(st->gp); free(st->pg); free(st->wp); free(st->d);
    if (st->s) for (size_t i = 0; i < st->vars.m; ++i) free(st->s[i]);
    if (st->y) for (size_t i = 0; i < st->vars.m; ++i) free(st->y[i]);
    free(st->s); free(st->y); free(st->ys); free(st->alpha);
    free(st);
}

int oracle_build(oracle_state** out, const oracle_param* p, double* x, size_t n,
                 oracle_eval_cb eval, void* user) {
    *out = NULL;
    oracle_state* st = (oracle_state*)calloc(1, sizeof(*st));
    if (!st) return ORACLE_ERR_ALLOC;
    st->vars = *p;
    st->n = n;
    st->x = x;
    st->eval = eval;
    st->user = user;
    size_t m = (size_t)p->m;
    /* lbfgs.rs:449, 620-627: m zeroed (s, y) pairs; core.rs:61-68: six zeroed vectors */
    st->s = (double**)calloc(m ? m : 1, sizeof(double*));
    st->y = (double**)calloc(m ? m : 1, sizeof(double*));
    st->ys = zalloc(m);
    st->alpha = zalloc(m);
    st->gx = zalloc(n); st->xp = zalloc(n); st->gp = zalloc(n);
    st->pg = zalloc(n); st->wp = zalloc(n); st->d = zalloc(n);
    int ok = st->s && st->y && st->ys && st->alpha && st->gx && st->xp && st->gp && st->pg &&
             st->wp && st->d;
    for (size_t i = 0; ok && i < m; ++i) {
        st->s[i] = zalloc(n);
        st->y[i] = zalloc(n);
        ok = st->s[i] && st->y[i];
    }
    if (!ok) { oracle_free(st); return ORACLE_ERR_ALLOC; }

    int rc = problem_evaluate(st); /* :454 */
    if (rc != ORACLE_OK) { oracle_free(st); return rc; }
    problem_update_search_direction(st); /* :457 */
    double h0 = p->initial_inverse_hessian;
    st->step = oracle_vec2norminv(st->d, n) * h0; /* :461 */
    st->end = 0;
    st->k = 0;
    st->ncall = 0;
    *out = st;
    return ORACLE_OK;
}

/* core.rs:253-268 */
static void get_progress(const oracle_state* st, oracle_progress* pr) {
    pr->x = st->x;
    pr->gx = st->gx;
    pr->fx = st->fx;
    pr->xnorm = problem_xnorm(st);
    pr->gnorm = problem_gnorm(st);
    pr->neval = st->neval;
    pr->ncall = st->ncall;
    pr->step = st->step;
    pr->niter = st->k;
}

/* lbfgs.rs:489-494, 697-748 */
int oracle_is_converged(oracle_state* st) {
    oracle_progress pr;
    get_progress(st, &pr);
    const oracle_param* p = &st->vars;
    if (p->max_iterations != 0 && pr.niter >= p->max_iterations) return 1;   /* :726-735 */
    if (p->max_evaluations != 0 && pr.neval >= p->max_evaluations) return 1; /* :739-748 */
    if (pr.gnorm / rmax(pr.xnorm, 1.0) <= p->epsilon) return 1;              /* :715 */
    return 0;
}

/* lbfgs.rs:503-560 */
int oracle_propagate(oracle_state* st, oracle_progress* out) {
    st->k += 1;
    if (st->k == 1) { /* :507-510 no-op first iteration */
        if (out) get_progress(st, out);
        return ORACLE_OK;
    }
    problem_save_state(st); /* :514 */

    uint64_t ncall = 0;
    int rc = oracle_line_search(st, &st->step, &ncall); /* :517-521 */
    if (rc != ORACLE_OK) return rc;
    st->ncall = ncall;
    double step_ls = st->step;

    size_t m = (size_t)st->vars.m, n = st->n;
    double gamma = 0.0;
    rc = oracle_history_update(st->s[st->end], st->y[st->end], &st->ys[st->end], &gamma, st->x,
                               st->xp, st->gx, st->gp, st->step, st->vars.damping, n, NULL); /* :525-533 */
    if (rc != ORACLE_OK) return rc;
    st->last_gamma = gamma;

    problem_update_search_direction(st); /* :536 */
    st->end = oracle_two_loop(st->s, st->y, st->ys, st->alpha, st->d, gamma, m,
                              (size_t)(st->k - 1), st->end, n); /* :540 */

    double dnorm = oracle_vec2norm(st->d, n); /* :543 */
    if (!sign_positive(dnorm)) return ORACLE_ERR_INVALID_DNORM;
    if (st->vars.constrain_step_size)
        st->step = rmin(st->vars.max_step_size, dnorm) / dnorm; /* :548 */
    else
        st->step = 1.0;

    if (st->vars.orthantwise) { /* :554, core.rs:213-217, orthantwise.rs:140-161 */
        oracle_project(st->d, st->pg, st->owl_start, st->owl_end, 1);
        if (oracle_vec2norm(st->d, n) == 0.0) return ORACLE_PANIC_ZERO_DIRECTION;
    }
    if (out) {
        get_progress(st, out);
        out->step = step_ls; /* :557 */
    }
    return ORACLE_OK;
}

/* lbfgs.rs:497-499, core.rs:288-298 */
void oracle_get_report(oracle_state* st, oracle_report* out) {
    out->fx = st->fx;
    out->xnorm = problem_xnorm(st);
    out->gnorm = problem_gnorm(st);
    out->neval = st->neval;
}

/* lbfgs.rs:399-421 */
int oracle_minimize(const oracle_param* p, double* x, size_t n, oracle_eval_cb eval, void* euser,
                    oracle_progress_cb progress, void* puser, oracle_report* out) {
    oracle_state* st = NULL;
    int rc = oracle_build(&st, p, x, n, eval, euser);
    if (rc != ORACLE_OK) return rc;
    for (;;) {
        if (oracle_is_converged(st)) break;
        oracle_progress pr;
        rc = oracle_propagate(st, &pr);
        if (rc != ORACLE_OK) { oracle_free(st); return rc; }
        if (progress && progress(puser, &pr)) break;
    }
    if (out) oracle_get_report(st, out);
    oracle_free(st);
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* introspection                                                               */
/* ------------------------------------------------------------------------- */
const double* oracle_state_vec(const oracle_state* st, int which) {
    switch (which) {
        case ORACLE_VEC_GX: return st->gx;
        case ORACLE_VEC_XP: return st->xp;
        case ORACLE_VEC_GP: return st->gp;
        case ORACLE_VEC_PG: return st->pg;
        case ORACLE_VEC_WP: return st->wp;
        case ORACLE_VEC_D: return st->d;
        case ORACLE_VEC_X: return st->x;
        default: return NULL;
    }
}
const double* oracle_state_hist(const oracle_state* st, size_t slot, int want_y) {
    if (slot >= st->vars.m) return NULL;
    return want_y ? st->y[slot] : st->s[slot];
}
double oracle_state_ys(const oracle_state* st, size_t slot) { return st->ys[slot]; }
double oracle_state_alpha(const oracle_state* st, size_t slot) { return st->alpha[slot]; }
size_t oracle_state_end(const oracle_state* st) { return st->end; }
size_t oracle_state_k(const oracle_state* st) { return (size_t)st->k; }
double oracle_state_step(const oracle_state* st) { return st->step; }
double oracle_state_gamma(const oracle_state* st) { return st->last_gamma; }
const char* oracle_state_ls_error(const oracle_state* st) { return st->ls_error; }
int oracle_state_ls_failed(const oracle_state* st) { return st->ls_failed; }

/* Which sources this library was built from: oracle.py passes the content hash of lbfgs_oracle.{c,h}, objectives.c and the
 * Makefile, and rebuilds when a prebuilt library carries another one (the string is readable from the file itself). */
#ifndef LBFGS_ORACLE_BUILD_ID
#define LBFGS_ORACLE_BUILD_ID "unstamped"
#endif
static const char oracle_build_id_marker[] = "LBFGS_ORACLE_BUILD_ID=" LBFGS_ORACLE_BUILD_ID;
const char* oracle_build_id(void) { return oracle_build_id_marker + sizeof("LBFGS_ORACLE_BUILD_ID=") - 1; }
