/*
 * oracle/objectives.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Test / benchmark objectives in the oracle's evaluate-callback form
 * (E: FnMut(&[f64], &mut [f64]) -> Result<f64>, reference src/core.rs:12).
 *
 *  - Rosenbrock: the reference's default_evaluate (src/lib.rs:79-94,
 *    benches/rosenbrock.rs:16-29).
 *  - Lennard-Jones all-pairs: the user objective of examples/lj.rs:20-64,113-118.
 *  - Hashed diagonal quadratic / separable logistic: the SYNTHETIC workloads of
 *    BASELINE.json configs 2-4 (not in the reference).  Their data is a
 *    counter-based hash of the GLOBAL element index, built from + and * only,
 *    so CPU, GPU and every shard generate bit-identical inputs with nothing to
 *    store.  The same formulas are restated in HIP in
 *    rust-lbfgs_amd/csrc/objectives.hip (product side, independent code).
 */
#include "lbfgs_oracle.h"

#include <math.h>
#include <stdlib.h>

/* f = sum of per-element (per-pair) terms.  Mode 0 (the reference): running sum in index order, exactly as the cited
 * loops.  Diagnostic modes 1 / 2: the same terms summed pairwise / from the last one down (oracle_sum_terms), so that the summation-order
 * sensitivity estimate of the tests also perturbs f, not only the dot products. */
typedef struct fsum {
    double acc;
    double* terms;
    size_t k;
} fsum;
static void fsum_begin(fsum* s, size_t nterms) {
    s->acc = 0.0;
    s->k = 0;
    s->terms = (oracle_dot_mode() != 0 && nterms) ? (double*)malloc(nterms * sizeof(double)) : NULL;
}
static inline void fsum_add(fsum* s, double t) {
    if (s->terms) s->terms[s->k++] = t;
    else s->acc += t;
}
static double fsum_end(fsum* s) {
    if (!s->terms) return s->acc;
    double r = oracle_sum_terms(s->terms, s->k);
    free(s->terms);
    return r;
}

/* src/lib.rs:79-94 */
double oracle_obj_rosenbrock(void* user, const double* x, double* g, size_t n, int* failed) {
    (void)user;
    (void)failed;
    fsum fx;
    fsum_begin(&fx, n / 2);
    for (size_t i = 0; i + 1 < n; i += 2) {
        double t1 = 1.0 - x[i];
        double t2 = 10.0 * (x[i + 1] - x[i] * x[i]);
        g[i + 1] = 20.0 * t2;
        g[i] = -2.0 * (x[i] * g[i + 1] + t1);
        fsum_add(&fx, t1 * t1 + t2 * t2);
    }
    return fsum_end(&fx);
}

/* splitmix64 finaliser applied to a counter: state = seed + (i+1)*golden */
static uint64_t mix64(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

double oracle_hash_u01(uint64_t seed, uint64_t i) {
    return (double)(mix64(seed, i) >> 11) * (1.0 / 9007199254740992.0);
}

/* f = sum_i x_i*(0.5*a_i*x_i - b_i),  g_i = a_i*x_i - b_i
 * a_i = 1 + 999*u_a^2 (condition number 1e3), b_i = 2*u_b - 1 */
double oracle_obj_quadratic(void* user, const double* x, double* g, size_t n, int* failed) {
    const oracle_hashed_obj* o = (const oracle_hashed_obj*)user;
    (void)failed;
    fsum fx;
    fsum_begin(&fx, n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t gi = o->global_offset + i;
        double ua = oracle_hash_u01(o->seed_a, gi);
        double ub = oracle_hash_u01(o->seed_b, gi);
        double a = 1.0 + 999.0 * (ua * ua);
        double b = 2.0 * ub - 1.0;
        double t = a * x[i];
        g[i] = t - b;
        fsum_add(&fx, x[i] * (0.5 * t - b));
    }
    return fsum_end(&fx);
}

/* f = sum_i log(1 + exp(-z_i)), z_i = w_i*x_i, w_i = t_i*a_i,
 * a_i = 0.5 + 1.5*u_a, t_i = +1 if top hash bit of (seed_b, i) set else -1
 * g_i = -w_i / (1 + exp(z_i)) */
double oracle_obj_logistic(void* user, const double* x, double* g, size_t n, int* failed) {
    const oracle_hashed_obj* o = (const oracle_hashed_obj*)user;
    (void)failed;
    fsum fx;
    fsum_begin(&fx, n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t gi = o->global_offset + i;
        double a = 0.5 + 1.5 * oracle_hash_u01(o->seed_a, gi);
        double w = (mix64(o->seed_b, gi) >> 63) ? a : -a;
        double z = w * x[i];
        double e = exp(-fabs(z));
        double fi = log1p(e);
        double sig; /* 1/(1+exp(z)) */
        if (z >= 0.0) {
            sig = e / (1.0 + e);
        } else {
            sig = 1.0 / (1.0 + e);
            fi -= z;
        }
        g[i] = -w * sig;
        fsum_add(&fx, fi);
    }
    return fsum_end(&fx);
}

/* examples/lj.rs:20-64 with epsilon = sigma = 1, then gx.vecscale(-1.0) (:116) */
double oracle_obj_lj(void* user, const double* x, double* g, size_t n, int* failed) {
    (void)user;
    (void)failed;
    size_t na = n / 3;
    double energy = 0.0;
    for (size_t i = 0; i < n; ++i) g[i] = 0.0;
    for (size_t i = 0; i < na; ++i) {
        for (size_t j = 0; j < i; ++j) {
            double dx = x[3 * i] - x[3 * j];
            double dy = x[3 * i + 1] - x[3 * j + 1];
            double dz = x[3 * i + 2] - x[3 * j + 2];
            double r = sqrt(dx * dx + dy * dy + dz * dz);
            double q = 1.0 / r;
            double s2 = q * q;
            double s6 = s2 * s2 * s2;
            energy += 4.0 * (s6 * s6 - s6);                 /* pair_energy  :22-25 */
            double gr = 24.0 * (s6 - 2.0 * (s6 * s6)) / r;  /* pair_gradient :28-32 */
            for (int k = 0; k < 3; ++k) {
                double dr = x[3 * j + k] - x[3 * i + k];
                g[3 * i + k] += 1.0 * gr * dr / r;
                g[3 * j + k] += -1.0 * gr * dr / r;
            }
        }
    }
    for (size_t i = 0; i < n; ++i) g[i] *= -1.0;
    return energy;
}

/* The substitute evaluator of BASELINE config 5 (not in the reference): the pair terms of examples/lj.rs over a
 * FIXED neighbour table (ELL, int32 [max_nbr][natoms], column-major, -1 = empty) with a cutoff rc and the energy
 * shifted by v(rc): E = sum_{i<j, r<rc} [v(r) - v(rc)].  Every pair is listed from both ends. */
double oracle_obj_lj_neighbors(void* user, const double* x, double* g, size_t n, int* failed) {
    const oracle_lj_nbr* o = (const oracle_lj_nbr*)user;
    (void)failed;
    const size_t na = n / 3;
    const double rc2 = o->cutoff * o->cutoff;
    const double i6 = 1.0 / (rc2 * rc2 * rc2);
    const double eshift = 4.0 * (i6 * i6 - i6);
    double energy = 0.0;
    for (size_t i = 0; i < na; ++i) {
        double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
        for (uint32_t k = 0; k < o->max_nbr; ++k) {
            const int32_t j = o->nbr[(size_t)k * na + i];
            if (j < 0) continue;
            const double dx = x[3 * i] - x[3 * (size_t)j], dy = x[3 * i + 1] - x[3 * (size_t)j + 1],
                         dz = x[3 * i + 2] - x[3 * (size_t)j + 2];
            const double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < rc2) {
                const double inv2 = 1.0 / r2;
                const double s6 = inv2 * inv2 * inv2;
                e += 4.0 * (s6 * s6 - s6);
                const double c = 24.0 * (s6 - 2.0 * (s6 * s6)) * inv2;
                fx += c * dx; fy += c * dy; fz += c * dz;
                e -= eshift;
            }
        }
        g[3 * i] = fx; g[3 * i + 1] = fy; g[3 * i + 2] = fz;
        energy += 0.5 * e;
    }
    return energy;
}

/* Cutoff + shift rule with a cell list rebuilt at every call (no Verlet skin: nothing is carried between calls).
 * Per atom i the pair terms are accumulated over the 27 surrounding cells; every pair is seen from both ends, as in
 * oracle_obj_lj_neighbors.  A non-finite coordinate makes the call fail (the closure's Err). */
double oracle_obj_lj_cells(void* user, const double* x, double* g, size_t n, int* failed) {
    const oracle_lj_cells* o = (const oracle_lj_cells*)user;
    const size_t na = n / 3;
    const double rc = o->cutoff, rc2 = rc * rc;
    const double i6 = 1.0 / (rc2 * rc2 * rc2);
    const double eshift = 4.0 * (i6 * i6 - i6);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < na; ++i)
        for (int k = 0; k < 3; ++k) {
            const double v = x[3 * i + k];
            if (!(v - v == 0.0)) { /* NaN or inf */
                if (failed) *failed = 1;
                return 0.0;
            }
            if (v < lo[k]) lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
    if (na == 0) return 0.0;
    /* cells of side >= rc; at most ~2*na cells */
    double side = rc;
    size_t dim[3];
    for (;;) {
        double total = 1.0;
        for (int k = 0; k < 3; ++k) {
            dim[k] = (size_t)floor((hi[k] - lo[k]) / side) + 1;
            total *= (double)dim[k];
        }
        if (total <= 2.0 * (double)na + 64.0) break;
        side *= 1.26;
    }
    const size_t ncell = dim[0] * dim[1] * dim[2];
    size_t* start = (size_t*)calloc(ncell + 1, sizeof(size_t));
    size_t* cell = (size_t*)malloc(na * sizeof(size_t));
    size_t* order = (size_t*)malloc(na * sizeof(size_t));
    if (!start || !cell || !order) {
        free(start); free(cell); free(order);
        if (failed) *failed = 1;
        return 0.0;
    }
    for (size_t i = 0; i < na; ++i) {
        size_t c[3];
        for (int k = 0; k < 3; ++k) {
            c[k] = (size_t)floor((x[3 * i + k] - lo[k]) / side);
            if (c[k] >= dim[k]) c[k] = dim[k] - 1;
        }
        cell[i] = (c[2] * dim[1] + c[1]) * dim[0] + c[0];
        start[cell[i] + 1] += 1;
    }
    for (size_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
    {   /* counting sort, stable: atoms of a cell in ascending index order */
        size_t* cur = (size_t*)malloc(ncell * sizeof(size_t));
        if (!cur) { free(start); free(cell); free(order); if (failed) *failed = 1; return 0.0; }
        for (size_t c = 0; c < ncell; ++c) cur[c] = start[c];
        for (size_t i = 0; i < na; ++i) order[cur[cell[i]]++] = i;
        free(cur);
    }
    double energy = 0.0;
    for (size_t i = 0; i < na; ++i) {
        const size_t cx = cell[i] % dim[0], cy = (cell[i] / dim[0]) % dim[1], cz = cell[i] / (dim[0] * dim[1]);
        double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const long long ax = (long long)cx + dx, ay = (long long)cy + dy, az = (long long)cz + dz;
                    if (ax < 0 || ay < 0 || az < 0 || ax >= (long long)dim[0] || ay >= (long long)dim[1] ||
                        az >= (long long)dim[2])
                        continue;
                    const size_t c = ((size_t)az * dim[1] + (size_t)ay) * dim[0] + (size_t)ax;
                    for (size_t t = start[c]; t < start[c + 1]; ++t) {
                        const size_t j = order[t];
                        if (j == i) continue;
                        const double ddx = x[3 * i] - x[3 * j], ddy = x[3 * i + 1] - x[3 * j + 1],
                                     ddz = x[3 * i + 2] - x[3 * j + 2];
                        const double r2 = ddx * ddx + ddy * ddy + ddz * ddz;
                        if (r2 < rc2) {
                            const double inv2 = 1.0 / r2;
                            const double s6 = inv2 * inv2 * inv2;
                            e += 4.0 * (s6 * s6 - s6);                             /* pair_energy  lj.rs:22-25 */
                            const double cc = 24.0 * (s6 - 2.0 * (s6 * s6)) * inv2; /* pair_gradient / r  lj.rs:28-32 */
                            fx += cc * ddx; fy += cc * ddy; fz += cc * ddz;
                            e -= eshift;
                        }
                    }
                }
        g[3 * i] = fx; g[3 * i + 1] = fy; g[3 * i + 2] = fz;
        energy += 0.5 * e;
    }
    free(start); free(cell); free(order);
    return energy;
}
