/*
 * tests/support/c_caller.c -- a plain C caller of the C-ABI (TEST PROGRAM).
 *
 * The Rust shim (integration/rust-shim) cannot be compiled in this image (no rustc), so this program is the nearest
 * executable stand-in for a non-Python caller: built by gcc against include/lbfgs_solver.h + include/lbfgs_hip.h,
 * linked with -llbfgs_solver -llbfgs_hip, it does what the reference's own tests do through the crate API:
 *
 *   1. tests/simple.rs:16-40   lbfgs().minimize(&mut x, default_evaluate(), ..) on Rosenbrock N = 100 from
 *                              (-1.2, 1.0) x 50, with a C callback as the `evaluate` closure
 *                              (src/lib.rs:79-94): fx ~ 0 and every x_i ~ 1 within 1e-4;
 *   2. tests/simple.rs:42-54   .with_orthantwise(1.0, 0, 99) continued from that x: fx ~ 43.5025,
 *                              x[0] ~ 0.25, x[1] ~ 0.0575 within 1e-4;
 *   3. tests/simple.rs:57-83   Booth's function from (-1.2, 1.0): x ~ (1, 3) within 1e-6, progress callback in C;
 *   4. src/math.rs:84-122      the LbfgsMath known answers through the 1:1 primitives (vecadd, vecdot, vecscale,
 *                              vecdiff, vecncpy, veccpy, vec2norm, vec2norminv) on device vectors.
 *
 * Exit status: 0 = all checks passed; 77 = no HIP device (the library refused to create a context: there is no CPU
 * fallback); 1 = a check failed.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "lbfgs_solver.h"

static int failures = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            ++failures;                                    \
            fprintf(stderr, "CHECK FAILED %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                  \
            fprintf(stderr, "\n");                         \
        }                                                  \
    } while (0)

/* src/lib.rs:79-94 default_evaluate */
static double rosenbrock(void* user, const double* x, double* g, uint64_t n, int* failed) {
    (void)failed;
    unsigned long* calls = (unsigned long*)user;
    if (calls) ++*calls;
    double fx = 0.0;
    for (uint64_t i = 0; i < n; i += 2) {
        const double t1 = 1.0 - x[i];
        const double t2 = 10.0 * (x[i + 1] - x[i] * x[i]);
        g[i + 1] = 20.0 * t2;
        g[i] = -2.0 * (x[i] * g[i + 1] + t1);
        fx += t1 * t1 + t2 * t2;
    }
    return fx;
}

/* tests/simple.rs:62-76 */
static double booth(void* user, const double* x, double* g, uint64_t n, int* failed) {
    (void)user; (void)n; (void)failed;
    const double x1 = x[0], x2 = x[1];
    const double t1 = x1 + 2.0 * x2 - 7.0, t2 = 2.0 * x1 + x2 - 5.0;
    g[0] = 2.0 * t1 + 4.0 * t2;
    g[1] = 4.0 * t1 + 2.0 * t2;
    return t1 * t1 + t2 * t2;
}

struct prog {
    unsigned long calls;
    uint64_t last_niter;
    double last_fx;
};
static int progress(void* user, const lbfgs_progress* p) {
    struct prog* pr = (struct prog*)user;
    pr->calls += 1;
    pr->last_niter = p->niter;
    pr->last_fx = p->fx;
    return 0; /* false = continue (lbfgs.rs:412-416) */
}

static lbfgs_evaluator host_eval(lbfgs_host_eval_cb cb, void* user) {
    lbfgs_evaluator ev;
    memset(&ev, 0, sizeof(ev));
    ev.kind = LBFGS_EVAL_HOST;
    ev.host = cb;
    ev.user = user;
    return ev;
}

static int near(double a, double b, double eps) { return fabs(a - b) <= eps; }

int main(void) {
    enum { N = 100 };
    char err[256] = "";
    lbfgs_hip_ctx* ctx = NULL;
    int rc = lbfgs_hip_ctx_create(&ctx, 0, N, NULL, NULL, NULL);
    if (rc == LBFGS_HIP_ERR_NO_DEVICE) {
        printf("c_caller: no HIP device: %s\n", lbfgs_hip_last_error(NULL));
        return 77;
    }
    if (rc != LBFGS_HIP_OK) {
        fprintf(stderr, "ctx_create: [%d] %s\n", rc, lbfgs_hip_last_error(NULL));
        return 1;
    }
    CHECK(lbfgs_hip_abi_version() == LBFGS_HIP_ABI_VERSION, "ABI version");

    /* 1. tests/simple.rs:16-40 */
    double x[N];
    for (int i = 0; i < N; i += 2) { x[i] = -1.2; x[i + 1] = 1.0; }
    lbfgs_param p;
    lbfgs_param_default(&p);
    unsigned long ncalls = 0;
    lbfgs_evaluator ev = host_eval(rosenbrock, &ncalls);
    struct prog pr = {0, 0, 0.0};
    lbfgs_report rep;
    rc = lbfgs_minimize(ctx, &p, x, &ev, progress, &pr, &rep, err, sizeof(err));
    CHECK(rc == LBFGS_OK, "rosenbrock minimize: [%d] %s", rc, err);
    CHECK(near(rep.fx, 0.0, 1e-4), "fx = %.17g", rep.fx);
    for (int i = 0; i < N; ++i) CHECK(near(x[i], 1.0, 1e-4), "x[%d] = %.17g", i, x[i]);
    CHECK(rep.neval == ncalls && pr.calls == pr.last_niter && pr.last_fx == rep.fx, "bookkeeping: neval %llu calls %lu",
          (unsigned long long)rep.neval, ncalls);
    printf("rosenbrock: fx = %.17g  x[0] = %.17g  niter = %llu  neval = %llu\n", rep.fx, x[0],
           (unsigned long long)pr.last_niter, (unsigned long long)rep.neval);

    /* 2. tests/simple.rs:42-54, continued from the converged x */
    lbfgs_param_default(&p);
    p.orthantwise = 1; p.owl_c = 1.0; p.owl_start = 0; p.owl_end = 99;
    rc = lbfgs_minimize(ctx, &p, x, &ev, NULL, NULL, &rep, err, sizeof(err));
    CHECK(rc == LBFGS_OK, "owlqn minimize: [%d] %s", rc, err);
    CHECK(near(rep.fx, 43.5025, 1e-4), "owlqn fx = %.17g", rep.fx);
    CHECK(near(x[0], 0.25, 1e-4) && near(x[1], 0.0575, 1e-4), "owlqn x = %.17g %.17g", x[0], x[1]);
    printf("owlqn:      fx = %.17g  x[0] = %.17g  x[1] = %.17g\n", rep.fx, x[0], x[1]);
    lbfgs_hip_ctx_destroy(ctx);

    /* 3. tests/simple.rs:57-83 */
    double xb[2] = {-1.2, 1.0};
    rc = lbfgs_hip_ctx_create(&ctx, 0, 2, NULL, NULL, NULL);
    CHECK(rc == LBFGS_HIP_OK, "ctx_create(2)");
    lbfgs_param_default(&p);
    ev = host_eval(booth, NULL);
    rc = lbfgs_minimize(ctx, &p, xb, &ev, progress, &pr, &rep, err, sizeof(err));
    CHECK(rc == LBFGS_OK, "booth minimize: [%d] %s", rc, err);
    CHECK(near(xb[0], 1.0, 1e-6) && near(xb[1], 3.0, 1e-6), "booth x = %.17g %.17g", xb[0], xb[1]);
    printf("booth:      x = (%.17g, %.17g)\n", xb[0], xb[1]);
    lbfgs_hip_ctx_destroy(ctx);

    /* 4. src/math.rs:84-122 */
    rc = lbfgs_hip_ctx_create(&ctx, 0, 3, NULL, NULL, NULL);
    CHECK(rc == LBFGS_HIP_OK, "ctx_create(3)");
    lbfgs_hip_vec *vx = NULL, *vy = NULL, *vz = NULL;
    CHECK(lbfgs_hip_vec_alloc(ctx, &vx) == 0 && lbfgs_hip_vec_alloc(ctx, &vy) == 0 && lbfgs_hip_vec_alloc(ctx, &vz) == 0,
          "vec_alloc");
    const double ones[3] = {1.0, 1.0, 1.0}, y0[3] = {1.0, 2.0, 3.0};
    double out[3], s = 0.0;
    lbfgs_hip_vec_upload(vx, ones, 3);
    lbfgs_hip_vec_upload(vy, y0, 3);
    lbfgs_hip_vecadd(vy, vx, 2.0);                       /* y += 2*x -> [3,4,5]      math.rs:91-93 */
    lbfgs_hip_vec_download(vy, out, 3);
    CHECK(out[0] == 3.0 && out[1] == 4.0 && out[2] == 5.0, "vecadd");
    lbfgs_hip_vecdot(vx, vy, 0);                         /* 12                       math.rs:96-97 */
    lbfgs_hip_scalars_read(ctx, 0, 1, &s);
    CHECK(s == 12.0, "vecdot = %g", s);
    lbfgs_hip_vecscale(vy, 2.0);                         /* [6,8,10]                 math.rs:100-102 */
    lbfgs_hip_vec_download(vy, out, 3);
    CHECK(out[0] == 6.0 && out[1] == 8.0 && out[2] == 10.0, "vecscale");
    lbfgs_hip_vecdiff(vz, vx, vy);                       /* x - y = [-5,-7,-9]       math.rs:105-110 */
    lbfgs_hip_vec_download(vz, out, 3);
    CHECK(out[0] == -5.0 && out[1] == -7.0 && out[2] == -9.0, "vecdiff");
    lbfgs_hip_vecncpy(vz, vx);                           /* [-1,-1,-1]               math.rs:118-121 */
    lbfgs_hip_vec_download(vz, out, 3);
    CHECK(out[0] == -1.0 && out[1] == -1.0 && out[2] == -1.0, "vecncpy");
    lbfgs_hip_veccpy(vz, vy);                            /* math.rs:113-115 */
    lbfgs_hip_vec_download(vz, out, 3);
    CHECK(out[0] == 6.0 && out[1] == 8.0 && out[2] == 10.0, "veccpy");
    double nrm = 0.0, inv = 0.0;
    CHECK(lbfgs_hip_vec2norm(vy, 1, &nrm) == 0 && nrm == sqrt(200.0), "vec2norm = %.17g", nrm);     /* math.rs:73-76 */
    CHECK(lbfgs_hip_vec2norminv(vy, 1, &inv) == 0 && inv == 1.0 / sqrt(200.0), "vec2norminv = %.17g", inv); /* :79-81 */
    printf("math:       vecadd vecdot vecscale vecdiff vecncpy veccpy vec2norm vec2norminv checked\n");
    lbfgs_hip_vec_free(vx);
    lbfgs_hip_vec_free(vy);
    lbfgs_hip_vec_free(vz);
    lbfgs_hip_ctx_destroy(ctx);

    if (failures) {
        fprintf(stderr, "c_caller: %d check(s) failed\n", failures);
        return 1;
    }
    printf("c_caller: all checks passed\n");
    return 0;
}
