// tools/bridge_rate.hip -- what the drop-in path costs: lbfgs().minimize(&mut x, evaluate, progress) with a HOST closure
// (lbfgs.rs:399-403, core.rs:119-121): x is downloaded and g uploaded around every evaluation (solver.cpp call_user_evaluate).
// Built by hipcc on the GPU box (tools/bridge_rate.sh); one JSON line per size.
//   1. the link: hipMemcpy between a hipHostMalloc'ed buffer and device memory, each way (best and median of 7) -- the rate the
//      staging is held against -- and the same from pageable memory, for comparison;
//   2. the library's staging calls themselves: lbfgs_hip_vec_download / lbfgs_hip_vec_upload on lbfgs_hip_host_buffer_create
//      buffers (what solver.cpp does per evaluation), same statistics;
//   3. lbfgs_minimize with a C closure that computes the diagonal quadratic in a tight loop (one host thread): per evaluation the
//      closure's own time, and per iteration the wall time -- so that  iteration = trials x (download + closure + upload) + the
//      GPU's vector work  can be checked from the record.
// usage: bridge_rate [n] [iterations]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lbfgs_solver.h"

namespace {
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Stat {
    double best_ms, median_ms;
};
template <class F>
Stat timed(F&& f, int reps = 7) {
    std::vector<double> t;
    f();  // warm-up (first touch, page tables)
    for (int i = 0; i < reps; ++i) {
        const double t0 = now();
        f();
        t.push_back((now() - t0) * 1e3);
    }
    std::sort(t.begin(), t.end());
    return Stat{t.front(), t[t.size() / 2]};
}

struct Closure {
    unsigned long calls = 0;
    double seconds = 0.0;
};
// f = sum 1/2 a_i x_i^2 - b_i x_i with a_i = 1 + (i mod 1000), b_i = 1 + (i mod 3): 1 read + 1 write of 8 bytes per element
double quadratic(void* user, const double* x, double* g, uint64_t n, int* failed) {
    (void)failed;
    Closure* c = static_cast<Closure*>(user);
    const double t0 = now();
    double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;  // four independent partial sums: the loop is not latency-bound on the adds
    uint64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        for (int k = 0; k < 4; ++k) {
            const uint64_t j = i + k;
            const double a = 1.0 + (double)(j % 1000), b = 1.0 + (double)(j % 3), xi = x[j];
            g[j] = a * xi - b;
            const double t = (0.5 * a * xi - b) * xi;
            if (k == 0) f0 += t; else if (k == 1) f1 += t; else if (k == 2) f2 += t; else f3 += t;
        }
    }
    for (; i < n; ++i) {
        const double a = 1.0 + (double)(i % 1000), b = 1.0 + (double)(i % 3), xi = x[i];
        g[i] = a * xi - b;
        f0 += (0.5 * a * xi - b) * xi;
    }
    c->calls += 1;
    c->seconds += now() - t0;
    return (f0 + f1) + (f2 + f3);
}

struct Rows {
    std::vector<double> t;
    std::vector<uint64_t> neval;
};
int progress(void* user, const lbfgs_progress* p) {
    Rows* r = static_cast<Rows*>(user);
    r->t.push_back(now());
    r->neval.push_back(p->neval);
    return 0;
}
}  // namespace

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint64_t iters = argc > 2 ? strtoull(argv[2], nullptr, 10) : 14;
    const size_t bytes = n * sizeof(double);
    lbfgs_hip_ctx* ctx = nullptr;
    int rc = lbfgs_hip_ctx_create(&ctx, 0, n, nullptr, nullptr, nullptr);
    if (rc != LBFGS_HIP_OK) {
        fprintf(stderr, "ctx_create: [%d] %s\n", rc, lbfgs_hip_last_error(nullptr));
        return rc == LBFGS_HIP_ERR_NO_DEVICE ? 77 : 1;
    }
    // ---- 1. the link
    void *dev = nullptr, *pin = nullptr;
    void* page = malloc(bytes);
    if (hipMalloc(&dev, bytes) != hipSuccess || hipHostMalloc(&pin, bytes, hipHostMallocDefault) != hipSuccess || !page) {
        fprintf(stderr, "allocation failed\n");
        return 1;
    }
    memset(pin, 0, bytes);
    memset(page, 0, bytes);
    const Stat d2h = timed([&] { (void)hipMemcpy(pin, dev, bytes, hipMemcpyDeviceToHost); });
    const Stat h2d = timed([&] { (void)hipMemcpy(dev, pin, bytes, hipMemcpyHostToDevice); });
    const Stat d2h_page = timed([&] { (void)hipMemcpy(page, dev, bytes, hipMemcpyDeviceToHost); }, 3);
    const Stat h2d_page = timed([&] { (void)hipMemcpy(dev, page, bytes, hipMemcpyHostToDevice); }, 3);
    (void)hipFree(dev);
    (void)hipHostFree(pin);
    free(page);
    // ---- 2. the library's staging calls
    lbfgs_hip_vec* v = nullptr;
    void* stage = nullptr;
    if (lbfgs_hip_vec_alloc(ctx, &v) != 0 || lbfgs_hip_host_buffer_create(ctx, bytes, &stage) != 0) {
        fprintf(stderr, "library allocation failed: %s\n", lbfgs_hip_last_error(ctx));
        return 1;
    }
    memset(stage, 0, bytes);
    const Stat down = timed([&] { (void)lbfgs_hip_vec_download(v, static_cast<double*>(stage), n); });
    const Stat up = timed([&] { (void)lbfgs_hip_vec_upload(v, static_cast<const double*>(stage), n); });
    lbfgs_hip_host_buffer_destroy(ctx, stage);
    lbfgs_hip_vec_free(v);
    // ---- 3. the drop-in path end to end
    lbfgs_param p;
    lbfgs_param_default(&p);
    p.m = 7;
    p.epsilon = 0.0;
    p.max_iterations = iters;
    Closure c;
    lbfgs_evaluator ev{};
    ev.kind = LBFGS_EVAL_HOST;
    ev.host = quadratic;
    ev.user = &c;
    std::vector<double> x(n, 0.0);
    Rows rows;
    lbfgs_report rep;
    char err[256] = "";
    rc = lbfgs_minimize(ctx, &p, x.data(), &ev, progress, &rows, &rep, err, sizeof(err));
    if (rc != LBFGS_OK || rows.t.size() < 6) {
        fprintf(stderr, "lbfgs_minimize: [%d] %s\n", rc, err);
        return 1;
    }
    // iterations 4.. (steady state): wall time and evaluations between two progress reports
    const size_t first = 3, last = rows.t.size() - 1;
    const double it_ms = (rows.t[last] - rows.t[first]) * 1e3 / (double)(last - first);
    const double evals_per_it = (double)(rows.neval[last] - rows.neval[first]) / (double)(last - first);
    const double closure_ms = c.seconds * 1e3 / (double)c.calls;
    const double gb = (double)bytes / 1e9;
    const double bridge_ms = evals_per_it * (down.median_ms + closure_ms + up.median_ms);
    printf("{\"n\": %llu, \"vector_MB\": %.1f, "
           "\"link_hipMemcpy_pinned\": {\"d2h_ms\": %.3f, \"d2h_GBps\": %.2f, \"h2d_ms\": %.3f, \"h2d_GBps\": %.2f, \"d2h_median_ms\": %.3f, \"h2d_median_ms\": %.3f}, "
           "\"link_hipMemcpy_pageable\": {\"d2h_ms\": %.3f, \"d2h_GBps\": %.2f, \"h2d_ms\": %.3f, \"h2d_GBps\": %.2f}, "
           "\"library_staging\": {\"download_ms\": %.3f, \"download_GBps\": %.2f, \"upload_ms\": %.3f, \"upload_GBps\": %.2f, "
           "\"download_median_ms\": %.3f, \"upload_median_ms\": %.3f, \"download_frac_of_link\": %.3f, \"upload_frac_of_link\": %.3f}, "
           "\"drop_in_minimize\": {\"iterations\": %zu, \"evaluations\": %lu, \"evaluations_per_iteration\": %.2f, \"closure_ms_per_evaluation\": %.3f, "
           "\"closure_GBps\": %.2f, \"iteration_ms\": %.3f, \"iters_per_sec\": %.2f, \"bridge_ms_per_iteration\": %.3f, "
           "\"gpu_and_rest_ms_per_iteration\": %.3f, \"final_fx\": %.17g}}\n",
           (unsigned long long)n, bytes / 1e6, d2h.best_ms, gb / d2h.best_ms * 1e3, h2d.best_ms, gb / h2d.best_ms * 1e3, d2h.median_ms,
           h2d.median_ms, d2h_page.best_ms, gb / d2h_page.best_ms * 1e3, h2d_page.best_ms, gb / h2d_page.best_ms * 1e3, down.best_ms,
           gb / down.best_ms * 1e3, up.best_ms, gb / up.best_ms * 1e3, down.median_ms, up.median_ms, d2h.best_ms / down.best_ms,
           h2d.best_ms / up.best_ms, rows.t.size(), c.calls, evals_per_it, closure_ms, 2.0 * gb / closure_ms * 1e3, it_ms, 1e3 / it_ms,
           bridge_ms, it_ms - bridge_ms, rep.fx);
    lbfgs_hip_ctx_destroy(ctx);
    return 0;
}
