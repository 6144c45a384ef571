// tests/support/cpp_api_tests.cpp -- the reference's own end-to-end tests, written against include/lbfgs.hpp (TEST PROGRAM).
//
// include/lbfgs.hpp mirrors the crate's public API for C++ callers; this program is what the crate's tests look like on it:
//   test_lbfgs_rosenbrock   tests/simple.rs:16-55   Rosenbrock N = 100 with default_evaluate / default_progress, then OWL-QN
//                                                   (.with_orthantwise(1.0, 0, 99)) continued from the converged x
//   test_lbfgs_booth        tests/simple.rs:57-83   Booth's function through a closure
//   test_owlqn              tests/owlqn.rs:6-63     Poisson regression, 500 x 21 (tests/golden/poisson_{x,y}.csv = the reference's
//                                                   tests/{x,y}.csv), with_orthantwise(1.0, 1, 21), epsilon 1e-4: fx = -42724.136705
//   test_state_api          src/lbfgs.rs:443-565    build / is_converged / propagate / report, Progress fields, cancel
//   test_errors             src/lbfgs.rs:195-361,454; src/line.rs:213-220; orthantwise.rs:64   assert!s, Err from the closure in
//                                                   build (propagates) and inside a line search (swallowed -> "x not changed")
// Linked against liblbfgs_solver + liblbfgs_hip (GPU) or the CPU test double of the C-ABI (logic check).
// usage: cpp_api_tests <dir with poisson_x.csv, poisson_y.csv>      exit: 0 ok, 77 no HIP device, 1 a check failed
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "lbfgs.hpp"

using liblbfgs::default_evaluate;
using liblbfgs::default_progress;
using liblbfgs::lbfgs;
using liblbfgs::Progress;

static int failures = 0;
#define CHECK(cond, ...)                                                 \
    do {                                                                 \
        if (!(cond)) {                                                   \
            ++failures;                                                  \
            std::fprintf(stderr, "CHECK FAILED %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);                           \
            std::fprintf(stderr, "\n");                                  \
        }                                                                \
    } while (0)
// approx::assert_relative_eq!(a, b, epsilon = e): |a - b| <= e (the absolute branch is the one these tests rely on)
static bool relative_eq(double a, double b, double eps) { return std::fabs(a - b) <= eps; }
static auto quiet = [](const Progress&) { return false; };

static void test_lbfgs_rosenbrock() {
    const std::size_t N = 100;
    std::vector<double> x(N);
    for (std::size_t i = 0; i < N; i += 2) {
        x[i] = -1.2;
        x[i + 1] = 1.0;
    }
    auto prb = lbfgs().minimize(x, default_evaluate(), quiet);
    CHECK(relative_eq(0.0, prb.fx, 1e-4), "fx = %.17g", prb.fx);
    for (std::size_t i = 0; i < N; ++i) CHECK(relative_eq(1.0, x[i], 1e-4), "x[%zu] = %.17g", i, x[i]);
    // OWL-QN, continued from the converged x
    prb = lbfgs().with_orthantwise(1.0, 0, 99).minimize(x, default_evaluate(), quiet);
    CHECK(relative_eq(43.5025, prb.fx, 1e-4), "owlqn fx = %.17g", prb.fx);
    CHECK(relative_eq(0.2500, x[0], 1e-4) && relative_eq(0.0575, x[1], 1e-4), "owlqn x = %.17g %.17g", x[0], x[1]);
    std::printf("test_lbfgs_rosenbrock: fx = %.17g x[0] = %.17g x[1] = %.17g\n", prb.fx, x[0], x[1]);
}

static void test_lbfgs_booth() {
    std::vector<double> x = {-1.2, 1.0};
    auto evaluate = [](const double* xs, double* gx, std::size_t) {
        const double x1 = xs[0], x2 = xs[1];
        const double fx = std::pow(x1 + 2.0 * x2 - 7.0, 2) + std::pow(2.0 * x1 + x2 - 5.0, 2);
        gx[0] = 10.0 * x1 + 8.0 * x2 - 34.0;
        gx[1] = 8.0 * x1 + 10.0 * x2 - 38.0;
        return fx;
    };
    (void)lbfgs().minimize(x, evaluate, default_progress());
    CHECK(relative_eq(x[0], 1.0, 1e-6) && relative_eq(x[1], 3.0, 1e-6), "booth x = %.17g %.17g", x[0], x[1]);
    std::printf("test_lbfgs_booth: x = (%.17g, %.17g)\n", x[0], x[1]);
}

// the fixtures as R wrote them: a header line, then rows `"index",v1,v2,...` -- the first field of every line is dropped
static std::vector<double> read_csv(const std::string& path) {
    std::vector<double> v;
    std::ifstream f(path);
    std::string line;
    bool header = true;
    while (std::getline(f, line)) {
        if (header) {
            header = false;
            continue;
        }
        std::istringstream is(line);
        std::string field;
        bool first = true;
        while (std::getline(is, field, ',')) {
            if (first) {
                first = false;
                continue;
            }
            v.push_back(std::stod(field));
        }
    }
    return v;
}

static void test_owlqn(const std::string& dir) {
    const std::size_t nrow = 500, ncol = 21;
    const std::vector<double> y = read_csv(dir + "/poisson_y.csv");
    const std::vector<double> xs = read_csv(dir + "/poisson_x.csv");  // row-major 500 x 21 (DMatrix::from_vec(21, 500, x).transpose())
    CHECK(y.size() == nrow && xs.size() == nrow * ncol, "fixtures: %zu %zu", y.size(), xs.size());
    if (y.size() != nrow || xs.size() != nrow * ncol) return;
    const double prec = 0.0;
    // likelihood <- -(sum(y * Xbeta - exp(Xbeta)) - .5 * sum(par^2 * prec));  gradient <- -(crossprod(X, y - exp(Xbeta)) - par * prec)
    auto evaluate = [&](const double* par, double* gx, std::size_t n) {
        double fx = 0.0;
        std::vector<double> t(nrow);
        for (std::size_t r = 0; r < nrow; ++r) {
            double xb = 0.0;
            for (std::size_t c = 0; c < ncol; ++c) xb += xs[r * ncol + c] * par[c];
            const double e = std::exp(xb);
            fx += y[r] * xb - e;
            t[r] = y[r] - e;
        }
        double reg = 0.0;
        for (std::size_t c = 0; c < n; ++c) reg += prec * par[c] * par[c];
        for (std::size_t c = 0; c < ncol; ++c) {
            double g = 0.0;
            for (std::size_t r = 0; r < nrow; ++r) g += xs[r * ncol + c] * t[r];
            gx[c] = -g + par[c] * prec;
        }
        return -fx + 0.5 * reg;
    };
    std::vector<double> xinit(ncol, 0.0);
    std::size_t last = 0;
    auto prb = lbfgs().with_orthantwise(1.0, 1, 21).with_epsilon(1e-4).minimize(xinit, evaluate, [&](const Progress& prgr) {
        last = prgr.niter;
        return false;
    });
    CHECK(relative_eq(-42724.136705, prb.fx, 1e-6), "owlqn poisson fx = %.10f", prb.fx);
    std::printf("test_owlqn: fx = %.6f after %zu iterations\n", prb.fx, last);
}

static void test_state_api() {
    std::vector<double> x(100);
    for (std::size_t i = 0; i < 100; i += 2) {
        x[i] = -1.2;
        x[i + 1] = 1.0;
    }
    std::size_t calls = 0;
    auto counted = [&calls, f = default_evaluate()](const double* xx, double* gx, std::size_t n) mutable {
        ++calls;
        return f(xx, gx, n);
    };
    auto state = lbfgs().with_max_iterations(5).build(x, counted);
    std::size_t iters = 0;
    double fx_prev = std::numeric_limits<double>::infinity();
    while (!state.is_converged()) {
        const Progress p = state.propagate();
        ++iters;
        CHECK(p.niter == iters && p.neval == calls, "niter %zu / %zu, neval %zu / %zu", p.niter, iters, p.neval, calls);
        CHECK(p.x() == x, "Progress.x is the caller's x");  // (the reference's state holds &mut x)
        CHECK(p.gx().size() == x.size() && p.fx <= fx_prev, "fx must not increase: %g -> %g", fx_prev, p.fx);
        fx_prev = p.fx;
    }
    const auto rep = state.report();
    CHECK(iters == 5 && rep.neval == calls && rep.fx == fx_prev, "report: %zu iterations, neval %zu vs %zu", iters, rep.neval, calls);
    // cancel from the progress closure after the third iteration (lbfgs.rs:412-416)
    std::vector<double> x2 = {-1.2, 1.0};
    std::size_t seen = 0;
    (void)lbfgs().minimize(x2, default_evaluate(), [&](const Progress& p) { return (seen = p.niter) >= 3; });
    CHECK(seen == 3, "cancelled at iteration %zu", seen);
    std::printf("test_state_api: 5 iterations, fx = %.6g; cancel at 3\n", rep.fx);
}

static void test_errors() {
    auto panics = [](auto&& f) {
        try {
            f();
        } catch (const liblbfgs::Panic&) {
            return true;
        } catch (...) {
        }
        return false;
    };
    CHECK(panics([] { lbfgs().with_epsilon(-1.0); }), "with_epsilon(-1) must panic");
    CHECK(panics([] { lbfgs().with_linesearch_gtol(1.5); }), "with_linesearch_gtol(1.5) must panic");
    CHECK(panics([] { lbfgs().with_linesearch_algorithm("Newton"); }), "unknown algorithm: unimplemented!()");
    CHECK(panics([] {  // orthantwise.rs:64: start >= end
        std::vector<double> x = {-1.2, 1.0};
        (void)lbfgs().with_orthantwise(1.0, 5, 3).minimize(x, default_evaluate(), quiet);
    }), "invalid orthantwise range must panic");
    // Err from the closure in build propagates as it is (lbfgs.rs:454)
    struct Refused : std::runtime_error {
        using std::runtime_error::runtime_error;
    };
    bool got = false;
    try {
        std::vector<double> x = {-1.2, 1.0};
        (void)lbfgs().minimize(x, [](const double*, double*, std::size_t) -> double { throw Refused("refused"); }, quiet);
    } catch (const Refused&) {
        got = true;
    } catch (...) {
    }
    CHECK(got, "the closure's own error must come back from build");
    // Err inside a line search is swallowed (revert, Ok(0)): the update then fails with "x not changed" (lbfgs.rs:646)
    int code = 0;
    std::size_t n = 0;
    std::vector<double> x = {-1.2, 1.0}, x0 = x;
    try {
        (void)lbfgs().minimize(x, [&n, f = default_evaluate()](const double* xx, double* gx, std::size_t k) mutable {
            if (++n == 2) throw Refused("refused");
            return f(xx, gx, k);
        }, quiet);
    } catch (const liblbfgs::Error& e) {
        code = e.code();
    }
    CHECK(code == LBFGS_ERR_X_NOT_CHANGED && x == x0, "swallowed line-search failure: code %d", code);
    std::printf("test_errors: setters panic, closure errors propagate / are swallowed as in the reference\n");
}

int main(int argc, char** argv) {
    try {
        std::vector<double> probe = {0.0, 0.0};
        (void)lbfgs().with_max_iterations(1).minimize(probe, default_evaluate(), quiet);
    } catch (const liblbfgs::Error& e) {
        if (e.code() == LBFGS_HIP_ERR_NO_DEVICE) {
            std::printf("cpp_api_tests: no HIP device: %s\n", e.what());
            return 77;
        }
        std::fprintf(stderr, "cpp_api_tests: [%d] %s\n", e.code(), e.what());
        return 1;
    }
    try {
        test_lbfgs_rosenbrock();
        test_lbfgs_booth();
        test_owlqn(argc > 1 ? argv[1] : "tests/golden");
        test_state_api();
        test_errors();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "cpp_api_tests: unexpected exception: %s\n", e.what());
        return 1;
    }
    if (failures) {
        std::fprintf(stderr, "cpp_api_tests: %d check(s) failed\n", failures);
        return 1;
    }
    std::printf("cpp_api_tests: all checks passed\n");
    return 0;
}
