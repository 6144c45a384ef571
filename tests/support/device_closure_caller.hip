// tests/support/device_closure_caller.hip -- a caller that OWNS HIP kernels (TEST PROGRAM, built by hipcc on the GPU box).
//
// The user-side counterpart of include/lbfgs_solver.h's LBFGS_EVAL_DEVICE: `evaluate` (core.rs:119-121) as the caller's own
// kernels on device pointers, x and g never leaving HBM -- and, since ABI version 4, the optional DEFERRED TRIAL POINTS:
//   device_probe   f(xp + t*d) and grad.d, nothing written        (what a line search reads from a trial: line.rs:283-288)
//   device_accept  x = xp + t*d, g = grad f(x)                    (once per search, for the accepted step)
// The program minimises  f(x) = sum_i 1/2 a_i x_i^2 - b_i x_i,  a_i = 1 + (i mod 1000) (condition number 1e3), b_i = 1 + (i mod 3)  three times
// through lbfgs_minimize -- full evaluations, probes + accept, probes only -- and checks that the three runs take the same
// discrete decisions (iterations, trials per search, evaluations) and agree per iteration to 1e-10 in f, ||g|| and the step
// (the closure's own g.d sum and the library's differ in summation order, so the last bits may differ; on the CPU test
// double, where both are sequential sums, the runs are bitwise equal: tests/test_host_logic_cpu.py).  It prints one JSON
// line with the wall time per iteration of each mode.   usage: device_closure_caller [n] [iterations]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lbfgs_solver.h"

namespace {
constexpr int BLOCK = 256;
constexpr int GRID = 1024;

__device__ __forceinline__ double coef_a(uint64_t i) { return 1.0 + (double)(i % 1000); }
__device__ __forceinline__ double coef_b(uint64_t i) { return 1.0 + (double)(i % 3); }

// workgroup sum in a fixed order; thread 0 gets the result
__device__ double block_sum(double v) {
    __shared__ double s[BLOCK];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int w = BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    const double r = s[0];
    __syncthreads();
    return r;
}

// MODE 0: evaluate at x (g written).  MODE 1: probe at xp + t*d (nothing written).  MODE 2: accept (x and g written).
template <int MODE>
__global__ __launch_bounds__(BLOCK) void quad_kernel(const double* __restrict__ xin, const double* __restrict__ d, double t,
                                                     double* __restrict__ xout, double* __restrict__ g, uint64_t n,
                                                     double* __restrict__ part) {
    double f = 0.0, dg = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)GRID * BLOCK) {
        double x = xin[i];
        if (MODE != 0) x = __dadd_rn(x, __dmul_rn(t, d[i]));  // core.rs:157-158: a multiply, then an add
        const double a = coef_a(i), b = coef_b(i);
        const double gi = __dadd_rn(__dmul_rn(a, x), -b);
        f += __dmul_rn(__dadd_rn(__dmul_rn(__dmul_rn(0.5, a), x), -b), x);
        if (MODE == 1) dg += __dmul_rn(gi, d[i]);
        if (MODE == 2) xout[i] = x;
        if (MODE != 1) g[i] = gi;
    }
    const double bf = block_sum(f);
    const double bd = MODE == 1 ? block_sum(dg) : 0.0;
    if (threadIdx.x == 0) {
        part[blockIdx.x] = bf;
        part[GRID + blockIdx.x] = bd;
    }
}
__global__ __launch_bounds__(64) void finish_kernel(const double* __restrict__ part, double* __restrict__ out) {
    if (threadIdx.x < 2) {  // the GRID partials in index order: the same bits every time
        double s = 0.0;
        for (int b = 0; b < GRID; ++b) s += part[threadIdx.x * GRID + b];
        out[threadIdx.x] = s;
    }
}

struct Closure {
    double* part = nullptr;   // 2 * GRID partials + 2 results
    double* host = nullptr;   // pinned, 2 doubles
    unsigned long evaluates = 0, probes = 0, accepts = 0;
};

void fetch(Closure* c, hipStream_t s) {
    finish_kernel<<<1, 64, 0, s>>>(c->part, c->part + 2 * GRID);
    (void)hipMemcpyAsync(c->host, c->part + 2 * GRID, 2 * sizeof(double), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
}

double evaluate_cb(void* user, const void* x, void* g, uint64_t n, void* stream, int* failed) {
    Closure* c = static_cast<Closure*>(user);
    c->evaluates += 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    quad_kernel<0><<<GRID, BLOCK, 0, s>>>(static_cast<const double*>(x), nullptr, 0.0, nullptr, static_cast<double*>(g), n, c->part);
    fetch(c, s);
    if (hipGetLastError() != hipSuccess) *failed = 1;
    return c->host[0];
}
double probe_cb(void* user, const void* xp, const void* d, double t, uint64_t n, void* stream, double* dg_out, int* failed) {
    Closure* c = static_cast<Closure*>(user);
    c->probes += 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    quad_kernel<1><<<GRID, BLOCK, 0, s>>>(static_cast<const double*>(xp), static_cast<const double*>(d), t, nullptr, nullptr, n,
                                          c->part);
    fetch(c, s);
    if (hipGetLastError() != hipSuccess) *failed = 1;
    *dg_out = c->host[1];
    return c->host[0];
}
double accept_cb(void* user, const void* xp, const void* d, double t, void* x, void* g, uint64_t n, void* stream, int* failed) {
    Closure* c = static_cast<Closure*>(user);
    c->accepts += 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    quad_kernel<2><<<GRID, BLOCK, 0, s>>>(static_cast<const double*>(xp), static_cast<const double*>(d), t, static_cast<double*>(x),
                                          static_cast<double*>(g), n, c->part);
    (void)hipStreamSynchronize(s);
    if (hipGetLastError() != hipSuccess) *failed = 1;
    return 0.0;
}

struct Row {
    double fx, gnorm, step;
    uint64_t niter, neval, ncall;
    double t;  // seconds on the steady clock when the iteration was reported
};
int progress_cb(void* user, const lbfgs_progress* p) {
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    static_cast<std::vector<Row>*>(user)->push_back(Row{p->fx, p->gnorm, p->step, p->niter, p->neval, p->ncall, t});
    return 0;
}
}  // namespace

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint64_t iters = argc > 2 ? strtoull(argv[2], nullptr, 10) : 40;
    lbfgs_hip_ctx* ctx = nullptr;
    int rc = lbfgs_hip_ctx_create(&ctx, 0, n, nullptr, nullptr, nullptr);
    if (rc == LBFGS_HIP_ERR_NO_DEVICE) {
        printf("device_closure_caller: no HIP device: %s\n", lbfgs_hip_last_error(nullptr));
        return 77;
    }
    if (rc != LBFGS_HIP_OK || lbfgs_hip_abi_version() != LBFGS_HIP_ABI_VERSION) {
        fprintf(stderr, "ctx_create / ABI: [%d] %s\n", rc, lbfgs_hip_last_error(nullptr));
        return 1;
    }
    Closure c;
    if (hipMalloc(&c.part, (2 * GRID + 2) * sizeof(double)) != hipSuccess || hipHostMalloc(&c.host, 2 * sizeof(double)) != hipSuccess) {
        fprintf(stderr, "allocation failed\n");
        return 1;
    }
    lbfgs_param p;
    lbfgs_param_default(&p);
    p.m = 7;
    p.epsilon = 0.0;
    p.max_iterations = iters;
    const char* names[3] = {"full_evaluations", "probes_and_accept", "probes_only"};
    std::vector<Row> rows[3];
    std::vector<double> x[3];
    double ms_per_iter[3];
    unsigned long counts[3][3];
    int failures = 0;
    for (int mode = 0; mode < 3; ++mode) {
        lbfgs_evaluator ev{};
        ev.kind = LBFGS_EVAL_DEVICE;
        ev.device = evaluate_cb;
        ev.user = &c;
        if (mode >= 1) ev.device_probe = probe_cb;
        if (mode == 1) ev.device_accept = accept_cb;
        c.evaluates = c.probes = c.accepts = 0;
        x[mode].assign(n, 0.0);
        lbfgs_report rep;
        char err[256] = "";
        rc = lbfgs_minimize(ctx, &p, x[mode].data(), &ev, progress_cb, &rows[mode], &rep, err, sizeof(err));
        if (rc != LBFGS_OK || rows[mode].size() < 12) {
            fprintf(stderr, "%s: lbfgs_minimize: [%d] %s (%zu iterations)\n", names[mode], rc, err, rows[mode].size());
            return 1;
        }
        // iterations 10.. (history full), from one progress report to the last: no upload of x0, no download of the result
        ms_per_iter[mode] = (rows[mode].back().t - rows[mode][9].t) * 1e3 / (double)(rows[mode].size() - 10);
        counts[mode][0] = c.evaluates; counts[mode][1] = c.probes; counts[mode][2] = c.accepts;
    }
    // the three runs: same discrete decisions, same numbers to 1e-10
    double worst = 0.0;
    for (int mode = 1; mode < 3; ++mode) {
        if (rows[mode].size() != rows[0].size()) {
            fprintf(stderr, "%s: %zu iterations, full run %zu\n", names[mode], rows[mode].size(), rows[0].size());
            return 1;
        }
        for (size_t i = 0; i < rows[0].size(); ++i) {
            const Row &a = rows[0][i], &b = rows[mode][i];
            if (a.niter != b.niter || a.neval != b.neval || a.ncall != b.ncall) {
                fprintf(stderr, "%s: iteration %zu: niter/neval/ncall %llu/%llu/%llu vs %llu/%llu/%llu\n", names[mode], i,
                        (unsigned long long)b.niter, (unsigned long long)b.neval, (unsigned long long)b.ncall,
                        (unsigned long long)a.niter, (unsigned long long)a.neval, (unsigned long long)a.ncall);
                ++failures;
            }
            worst = fmax(worst, fabs(a.fx - b.fx) / fabs(a.fx));
            worst = fmax(worst, fabs(a.gnorm - b.gnorm) / a.gnorm);
            worst = fmax(worst, fabs(a.step - b.step) / fabs(a.step));
        }
        double xmax = 0.0, xdiff = 0.0;
        for (uint64_t i = 0; i < n; ++i) {
            xmax = fmax(xmax, fabs(x[0][i]));
            xdiff = fmax(xdiff, fabs(x[0][i] - x[mode][i]));
        }
        worst = fmax(worst, xdiff / xmax);
    }
    const uint64_t neval = rows[0].back().neval, searches = rows[0].size() - 1;
    // full: every evaluation is the closure's evaluate.  Deferred: one evaluate in build (lbfgs.rs:454), every trial a probe,
    // one accept -- or, without accept, one more evaluate -- per search
    if (counts[0][0] != neval || counts[0][1] != 0 || counts[0][2] != 0) ++failures;
    if (counts[1][0] != 1 || counts[1][1] != neval - 1 || counts[1][2] != searches) ++failures;
    if (counts[2][0] != 1 + searches || counts[2][1] != neval - 1 || counts[2][2] != 0) ++failures;
    if (!(worst <= 1e-10)) ++failures;
    printf("{\"n\": %llu, \"m\": 7, \"iterations\": %zu, \"neval\": %llu, \"trials_per_search\": %.3f, \"worst_relative_deviation\": %.3e, "
           "\"ms_per_iteration\": {\"%s\": %.4f, \"%s\": %.4f, \"%s\": %.4f}, "
           "\"closure_calls\": {\"%s\": [%lu, %lu, %lu], \"%s\": [%lu, %lu, %lu], \"%s\": [%lu, %lu, %lu]}, \"final_fx\": %.17g}\n",
           (unsigned long long)n, rows[0].size(), (unsigned long long)neval, (double)(neval - 1) / (double)searches, worst, names[0],
           ms_per_iter[0], names[1], ms_per_iter[1], names[2], ms_per_iter[2], names[0], counts[0][0], counts[0][1], counts[0][2],
           names[1], counts[1][0], counts[1][1], counts[1][2], names[2], counts[2][0], counts[2][1], counts[2][2], rows[0].back().fx);
    (void)hipFree(c.part);
    (void)hipHostFree(c.host);
    lbfgs_hip_ctx_destroy(ctx);
    if (failures) {
        fprintf(stderr, "device_closure_caller: %d check(s) failed\n", failures);
        return 1;
    }
    printf("device_closure_caller: all checks passed\n");
    return 0;
}
