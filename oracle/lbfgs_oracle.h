/*
 * oracle/lbfgs_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the arithmetic of the reference crate `liblbfgs`
 * (ybyygu/rust-lbfgs v0.2.0) for the hot path named by BASELINE.json: the
 * BLAS-1 primitives (src/math.rs), the Problem state (src/core.rs), the
 * L-BFGS iteration + two-loop recursion + history update (src/lbfgs.rs), the
 * line searches (src/line.rs) and the OWL-QN operators (src/orthantwise.rs).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (rust-lbfgs_amd/) never links or calls it.
 *
 * Parity pin: the Rust reference cannot be compiled in this environment (no
 * rustc/cargo), so this restatement is pinned by the reference's own fixtures:
 *   - src/math.rs:84-122      exact primitive results,
 *   - tests/simple.rs:33-35   Rosenbrock N=100 17-digit final state (comment),
 *   - tests/simple.rs:48-50   OWL-QN continuation 17-digit final state,
 *   - tests/simple.rs:37-54,81-82 and tests/owlqn.rs:60 assertions,
 * see tests/test_oracle_golden.py.  Paths the reference's tests never touch
 * (Powell damping, gradient_only, non-OWL-QN backtracking, m != 6) are
 * "parity unpinned": they follow the source text only.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA, no reassociation) -- see Makefile.
 */
#ifndef LBFGS_ORACLE_H
#define LBFGS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* line.rs:39-81 LineSearchAlgorithm */
enum {
    ORACLE_LS_MORETHUENTE = 0,
    ORACLE_LS_BACKTRACKING_ARMIJO = 1,
    ORACLE_LS_BACKTRACKING_STRONGWOLFE = 2,
    ORACLE_LS_BACKTRACKING_WOLFE = 3
};

/* status codes: 0 = Ok; <0 = the reference returns Err / panics at the cited site */
enum {
    ORACLE_OK = 0,
    ORACLE_ERR_EVALUATE = -1,          /* user evaluate returned Err (lbfgs.rs:454)          */
    ORACLE_ERR_NEGATIVE_STEP = -2,     /* line.rs:198-201                                    */
    ORACLE_ERR_GRADONLY_MT = -3,       /* line.rs:208                                        */
    ORACLE_ERR_X_NOT_CHANGED = -4,     /* lbfgs.rs:646                                       */
    ORACLE_ERR_GX_NOT_CHANGED = -5,    /* lbfgs.rs:655                                       */
    ORACLE_ERR_INVALID_DNORM = -6,     /* lbfgs.rs:544                                       */
    ORACLE_PANIC_OWLQN_RANGE = -20,    /* orthantwise.rs:64 assert!(start < end)             */
    ORACLE_PANIC_ZERO_DIRECTION = -21, /* orthantwise.rs:160 assert_ne!(||d||, 0)            */
    ORACLE_ERR_ALLOC = -30
};

/* LbfgsParam (lbfgs.rs:72-154) + LineSearch (line.rs:91-148) + Orthantwise (orthantwise.rs:19-45) */
typedef struct oracle_param {
    uint64_t m;                 /* lbfgs.rs:163 default 6 (no setter in the reference API) */
    double epsilon;             /* 1e-5 */
    uint64_t past;              /* 0   (dead: lbfgs.rs:702 commented out) */
    double delta;               /* 1e-5 (dead) */
    uint64_t max_iterations;    /* 0 */
    uint64_t max_evaluations;   /* 0 */
    double initial_inverse_hessian; /* 1.0 */
    double max_step_size;       /* 1.0 */
    int32_t damping;            /* false */
    int32_t constrain_step_size;/* true (no setter) */
    /* line search */
    int32_t ls_algorithm;       /* MoreThuente */
    int32_t gradient_only;      /* false */
    double ftol;                /* 1e-4 */
    double gtol;                /* 0.9 */
    double xtol;                /* f64::EPSILON */
    double min_step;            /* 1e-20 */
    double max_step;            /* 1e20 */
    uint64_t max_linesearch;    /* 20 */
    /* OWL-QN */
    int32_t orthantwise;        /* 0 = None */
    int32_t _pad;
    double owl_c;               /* 1.0 */
    uint64_t owl_start;         /* 0 */
    int64_t owl_end;            /* -1 = None (=> n) */
} oracle_param;

void oracle_param_default(oracle_param* p);

/* E: FnMut(&[f64], &mut [f64]) -> Result<f64>   (core.rs:12).  *failed != 0 <=> Err */
typedef double (*oracle_eval_cb)(void* user, const double* x, double* g, size_t n, int* failed);

/* Progress (core.rs:223-250) */
typedef struct oracle_progress {
    const double* x;
    const double* gx;
    double fx, xnorm, gnorm, step;
    uint64_t niter, neval, ncall;
} oracle_progress;

/* G: FnMut(&Progress) -> bool, true = cancel (lbfgs.rs:402,412-416) */
typedef int (*oracle_progress_cb)(void* user, const oracle_progress* p);

/* Report (core.rs:273-285) */
typedef struct oracle_report {
    double fx, xnorm, gnorm;
    uint64_t neval;
} oracle_report;

/* ---- math.rs:31-82 -------------------------------------------------------- */
/* dot mode: 0 = sequential left-to-right (the reference, math.rs:41);
 * 1 = pairwise tree, 2 = sequential from the LAST term down, 3 / 4 = 4 / 64 interleaved running sums joined by a tree
 * (diagnostic only: independent perturbations of the summation order, from which the tests estimate how strongly a run
 * amplifies last-bit differences; a run whose terms are all equal -- Rosenbrock from its standard start -- is blind to
 * mode 2).  Modes 1-4 reorder EVERY sum of the restatement: vecdot, x1norm and the f of the built-in objectives.
 * Process-global, default 0. */
void oracle_set_dot_mode(int mode);
int oracle_dot_mode(void);
/* sum of a term array in the current mode (sequential = the reference; pairwise = diagnostic) */
double oracle_sum_terms(const double* t, size_t n);
void oracle_vecadd(double* y, const double* x, double c, size_t n);    /* y += c*x  */
double oracle_vecdot(const double* x, const double* y, size_t n);      /* sum x*y   */
void oracle_vecscale(double* y, double c, size_t n);                   /* y *= c    */
void oracle_veccpy(double* y, const double* x, size_t n);              /* y = x     */
void oracle_vecncpy(double* y, const double* x, size_t n);             /* y = -x    */
void oracle_vecdiff(double* z, const double* x, const double* y, size_t n); /* z = x-y */
double oracle_vec2norm(const double* x, size_t n);
double oracle_vec2norminv(const double* x, size_t n);

/* ---- orthantwise.rs -------------------------------------------------------- */
double oracle_signum(double x);                                         /* :174-180 */
int oracle_owl_range(const oracle_param* p, size_t n, size_t* start, size_t* end); /* :59-67 */
double oracle_x1norm(double c, size_t start, size_t end, const double* x);         /* :70-79 */
void oracle_pseudo_gradient(double c, size_t start, size_t end, double* pg,
                            const double* x, const double* g, size_t n);           /* :82-112 */
void oracle_project(double* x, const double* w, size_t start, size_t end, int negate_w); /* :165-171 */
void oracle_orthant_select(double* wp, const double* xp, const double* pg, size_t n);   /* core.rs:167-180 */

/* ---- lbfgs.rs:569-604 two-loop on caller-owned history ---------------------- */
/* s, y: m pointers to n-vectors; ys, alpha: m scalars (alpha is overwritten).
 * Returns the new `end`.  d is updated in place. */
size_t oracle_two_loop(double* const* s, double* const* y, const double* ys, double* alpha,
                       double* d, double gamma, size_t m, size_t k, size_t end, size_t n);

/* ---- lbfgs.rs:640-692 IterationData::update -------------------------------- */
/* Writes s, y, *ys; returns status; *gamma = ys/yy.  aux[0..4] (optional):
 * ||s||, ys, yy, sbs, damping case (0 none/3, 1, 2). */
int oracle_history_update(double* s, double* y, double* ys_out, double* gamma,
                          const double* x, const double* xp, const double* gx, const double* gp,
                          double step, int damping, size_t n, double* aux);

/* ---- LbfgsState (lbfgs.rs:425-566) ------------------------------------------ */
typedef struct oracle_state oracle_state;

int oracle_build(oracle_state** out, const oracle_param* p, double* x, size_t n,
                 oracle_eval_cb eval, void* user);                       /* lbfgs.rs:443-481 */
int oracle_is_converged(oracle_state* st);                               /* lbfgs.rs:489-494 */
int oracle_propagate(oracle_state* st, oracle_progress* out);            /* lbfgs.rs:503-560 */
void oracle_get_report(oracle_state* st, oracle_report* out);            /* lbfgs.rs:497-499 */
void oracle_free(oracle_state* st);

/* minimize (lbfgs.rs:399-421).  progress may be NULL (never cancels). */
int oracle_minimize(const oracle_param* p, double* x, size_t n, oracle_eval_cb eval, void* euser,
                    oracle_progress_cb progress, void* puser, oracle_report* out);

/* introspection for step-locked parity tests */
enum { ORACLE_VEC_GX = 0, ORACLE_VEC_XP, ORACLE_VEC_GP, ORACLE_VEC_PG, ORACLE_VEC_WP, ORACLE_VEC_D,
       ORACLE_VEC_X };
const double* oracle_state_vec(const oracle_state* st, int which);
const double* oracle_state_hist(const oracle_state* st, size_t slot, int want_y);
double oracle_state_ys(const oracle_state* st, size_t slot);
double oracle_state_alpha(const oracle_state* st, size_t slot);
size_t oracle_state_end(const oracle_state* st);
size_t oracle_state_k(const oracle_state* st);
double oracle_state_step(const oracle_state* st);
double oracle_state_gamma(const oracle_state* st);   /* last gamma = ys/yy */
/* message of the last swallowed line-search failure (line.rs:213-220), "" if none */
const char* oracle_state_ls_error(const oracle_state* st);
const char* oracle_build_id(void); /* content hash of the sources this library was built from (oracle.py) */
/* 1 if the last line search failed and was swallowed (quirk 6) */
int oracle_state_ls_failed(const oracle_state* st);

/* one stand-alone line search on a built state (line.rs:193-223); returns status, *ncall */
int oracle_line_search(oracle_state* st, double* step, uint64_t* ncall);

/* ---- synthetic / test objectives (oracle/objectives.c) ----------------------- */
/* All are oracle_eval_cb-compatible; `user` points at the struct shown. */
double oracle_obj_rosenbrock(void* user, const double* x, double* g, size_t n, int* failed); /* lib.rs:79-94 */

typedef struct oracle_hashed_obj {
    uint64_t global_offset;   /* index of local element 0 in the global vector */
    uint64_t seed_a, seed_b;
} oracle_hashed_obj;
/* u(i; seed) = top 53 bits of splitmix64 counter hash / 2^53, in [0,1) */
double oracle_hash_u01(uint64_t seed, uint64_t i);
/* f = sum 0.5*a_i x_i^2 - b_i x_i ; a_i = 1 + 999 u_a^2 ; b_i = 2 u_b - 1 */
double oracle_obj_quadratic(void* user, const double* x, double* g, size_t n, int* failed);
/* f = sum log(1+exp(-t_i a_i x_i)) ; a_i = 0.5 + 1.5 u_a ; t_i = +-1 from hash bit of seed_b */
double oracle_obj_logistic(void* user, const double* x, double* g, size_t n, int* failed);
/* Lennard-Jones all-pairs, x = 3*natoms coordinates; returns energy, g = dE/dx (examples/lj.rs:38-64,113-118) */
double oracle_obj_lj(void* user, const double* x, double* g, size_t n, int* failed);
/* the same pair terms over a fixed neighbour table with a cutoff (substitute evaluator, config 5) */
typedef struct oracle_lj_nbr {
    const int32_t* nbr;   /* int32 [max_nbr][natoms], column-major, -1 = empty */
    uint32_t max_nbr;
    uint32_t _pad;
    double cutoff;
} oracle_lj_nbr;
double oracle_obj_lj_neighbors(void* user, const double* x, double* g, size_t n, int* failed);
/* the truncated-and-shifted pair potential of the substitute evaluator with NO stored list:
 *   E = sum_{i<j, r<rc} [v(r) - v(rc)],  g = dE/dx,
 * found through a cell list that is rebuilt at EVERY call (cells of side >= rc), so the result depends on x alone.
 * This is the rule the device's rebuildable Verlet list (LBFGS_HIP_OBJ_LJ_CELLS: cutoff + skin, rebuilt when an atom
 * has moved more than skin/2) must reproduce at every point, whatever its rebuild history. */
typedef struct oracle_lj_cells {
    double cutoff;
} oracle_lj_cells;
double oracle_obj_lj_cells(void* user, const double* x, double* g, size_t n, int* failed);

#ifdef __cplusplus
}
#endif
#endif
