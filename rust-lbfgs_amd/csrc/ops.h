// rust-lbfgs_amd/csrc/ops.h -- element operators of the L-BFGS / OWL-QN hot path.
//
// Each operator states what ONE element does; stream.h turns it into a
// bandwidth-bound kernel.  Arithmetic keeps the reference's roundings: the
// translation unit is compiled with -ffp-contract=off, so `y + c*x` is a
// multiply then an add exactly as in the reference's src/math.rs:35.
// Citations are file:line in the reference crate (ybyygu/rust-lbfgs).
#pragma once
#include "stream.h"

namespace lh {

struct NoCoef {};

// orthantwise.rs:174-180: NaN and +-0 -> 0, else the sign
__device__ __forceinline__ double signum0(double x) {
    if (x != x || x == 0.0) return 0.0;
    return (__double_as_longlong(x) < 0) ? -1.0 : 1.0;
}

// ---------------------------------------------------------------- math.rs primitives
struct OpDot {  // math.rs:40-42
    static constexpr int NIN = 2, NOUT = 0, NRED = 1;
    const double* in[2];
    double* out[1];
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double*, double* acc, uint64_t) const { acc[0] += v[0] * v[1]; }
};

struct OpNrm2 {  // math.rs:73-76 (squared)
    static constexpr int NIN = 1, NOUT = 0, NRED = 1;
    const double* in[1];
    double* out[1];
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double*, double* acc, uint64_t) const { acc[0] += v[0] * v[0]; }
};

struct OpNorms2 {  // core.rs:261-262 ||x||^2 and ||g||^2 in one pass
    static constexpr int NIN = 2, NOUT = 0, NRED = 2;
    const double* in[2];
    double* out[1];
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double*, double* acc, uint64_t) const {
        acc[0] += v[0] * v[0];
        acc[1] += v[1] * v[1];
    }
};

struct OpAxpy {  // math.rs:33-37  y += c*x ; c from the host or from the board
    static constexpr int NIN = 2, NOUT = 1, NRED = 0;
    const double* in[2];  // y, x
    double* out[1];       // y
    double c_host;
    const double* c_dev;  // nullable
    typedef double Coef;
    __device__ Coef setup() const { return c_dev ? *c_dev : c_host; }
    __device__ void elem(const Coef& c, const double* v, double* w, double*, uint64_t) const { w[0] = v[0] + c * v[1]; }
};

struct OpScale {  // math.rs:45-49
    static constexpr int NIN = 1, NOUT = 1, NRED = 0;
    const double* in[1];
    double* out[1];
    double c;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double*, uint64_t) const { w[0] = v[0] * c; }
};

template <bool NEG>
struct OpCopy {  // math.rs:52-56 / 59-63
    static constexpr int NIN = 1, NOUT = 1, NRED = 0;
    const double* in[1];
    double* out[1];
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double*, uint64_t) const { w[0] = NEG ? -v[0] : v[0]; }
};

struct OpDiff {  // math.rs:66-70  z = x - y
    static constexpr int NIN = 2, NOUT = 1, NRED = 0;
    const double* in[2];
    double* out[1];
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double*, uint64_t) const { w[0] = v[0] - v[1]; }
};

// ---------------------------------------------------------------- line step
// core.rs:155-164: x = xp (veccpy) ; x += step*d (vecadd) ; [project onto wp on [start,end)]
template <bool PROJECT>
struct OpLineStep {
    static constexpr int NIN = PROJECT ? 3 : 2, NOUT = 1, NRED = 0;
    const double* in[3];  // xp, d, wp
    double* out[1];       // x
    double step;
    uint64_t start, end;  // global
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double*, uint64_t gi) const {
        double x = v[0] + step * v[1];
        if constexpr (PROJECT) {
            // orthantwise.rs:165-171
            if (gi >= start && gi < end && signum0(x) != signum0(v[2])) x = 0.0;
        }
        w[0] = x;
    }
};

// ---------------------------------------------------------------- history update
// lbfgs.rs:640-673, one pass: s = x-xp, y = g-gp, ||s||^2, y.s, y.y, ||x||^2, ||g||^2, s.bs (0 unless DAMP),
// and s.(-g): the numerator of the two-loop's FIRST alpha (lbfgs.rs:587 with d = -g, core.rs:95-101), which
// needs exactly the two vectors this kernel already holds.
template <bool DAMP>
struct OpHistUpdate {
    static constexpr int NIN = 4, NOUT = 2, NRED = 7;
    const double* in[4];  // x, xp, g, gp
    double* out[2];       // s, y
    double neg_step;      // -step (lbfgs.rs:671)
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t) const {
        const double s = v[0] - v[1];
        const double y = v[2] - v[3];
        w[0] = s;
        w[1] = y;
        acc[0] += s * s;
        acc[1] += y * s;
        acc[2] += y * y;
        acc[3] += v[0] * v[0];
        acc[4] += v[2] * v[2];
        if constexpr (DAMP) acc[5] += s * (v[3] * neg_step);
        acc[6] += s * (-v[2]);
    }
};

// lbfgs.rs:675-680 damping case 1: bs = gp*(-step); bs *= (1-theta); bs += theta*y; y = bs
struct OpDamp {
    static constexpr int NIN = 2, NOUT = 1, NRED = 0;
    const double* in[2];  // gp, y
    double* out[1];       // y
    double neg_step, one_minus_theta, theta;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double*, uint64_t) const {
        double bs = v[0] * neg_step;
        bs = bs * one_minus_theta;
        w[0] = bs + theta * v[1];
    }
};

// ---------------------------------------------------------------- two-loop recursion (lbfgs.rs:569-604)
// First dot: alpha_0 numerator = s . d with d = -g folded in (core.rs:95-101).  2r.
struct OpTwoLoopFirst {
    static constexpr int NIN = 2, NOUT = 0, NRED = 1;
    const double* in[2];  // g, s
    double* out[1];
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double*, double* acc, uint64_t) const { acc[0] += v[1] * (-v[0]); }
};

// One fused step:   q = [ -src | src ] + c*u ;  [ q *= gamma ] ;  dst = q ;  sum += w*q
//   VMODE 0: w is a third stream v        (3r 1w)  -- the dominant kernel
//   VMODE 1: w = u  (gamma transition)    (2r 1w)
//   VMODE 2: w = q  (last step): sums ||d||^2 AND g.d, the next line search's dginit
//            (core.rs:78-92), reading g as the third stream                      (3r 1w)
//   VMODE 3: VMODE 2 under OWL-QN: the third stream is pg; ||d||^2 is summed BEFORE the orthant projection
//            (lbfgs.rs:543 precedes :554), then d_i = 0 where signum(d_i) != signum(-pg_i) on [start,end)
//            (orthantwise.rs:140-161), and ||d||^2, pg.d of the projected direction are summed too   (3r 1w)
// The coefficient is formed on the device from the previous reduction (no host round trip):
//   mode A (first loop, lbfgs.rs:587-589):  alpha_j = dot/ys_j ; c = -alpha_j   (alpha_j is stored)
//   mode B (second loop, lbfgs.rs:597-599): beta = dot/ys_j ;    c = alpha_j - beta
struct TwoLoopCoef {
    double c, gamma;
};
#ifndef LH_STEP_MAP      // overridable for in-situ A/B builds (tools/build_variants.sh); the defaults are the measured best
#define LH_STEP_MAP 1
#endif
#ifndef LH_STEP_UNROLL
#define LH_STEP_UNROLL 2
#endif
template <bool NEG_SRC, bool SCALE, int VMODE>
struct OpTwoLoopStep {
    static constexpr int NIN = (VMODE == 1) ? 2 : 3, NOUT = 1, NRED = (VMODE == 2) ? 2 : (VMODE == 3 ? 4 : 1);
    // measured best for the 3r+1w shape on MI355X: fine grid-stride, 2 chunks in flight per stream
    static constexpr int TUNE_MAP = (VMODE == 1) ? DEFAULT_MAP : LH_STEP_MAP, TUNE_UNROLL = (VMODE == 1) ? UNROLL : LH_STEP_UNROLL;  // 3r+1w shapes
    const double* in[3];  // src, u, v (VMODE 2: g)
    double* out[1];       // dst (= d)
    const double* dot_in; // previous reduction (global sum) ...
    const double* dot_parts;  // ... or, when dot_nparts > 0, the producing kernel's dot_nparts workgroup partials, which
    unsigned int dot_nparts;  //     this kernel adds up itself (stream.h sum_partials: the reducer's order, bitwise)
    const double* ys_j;   // ys of the slot whose coefficient this step applies
    double* alpha_j;      // alpha of that slot (written in mode A, read in mode B)
    const double* gamma_num;
    const double* gamma_den;
    int mode_b;
    uint64_t owl_start, owl_end;  // VMODE 3 only
    typedef TwoLoopCoef Coef;
    __device__ Coef setup() const {
        Coef cf;
        const double dot = dot_nparts ? sum_partials(dot_parts, dot_nparts) : *dot_in;
        const double r = dot / *ys_j;
        if (mode_b) {
            cf.c = *alpha_j - r;
        } else {
            cf.c = -r;
            if (blockIdx.x == 0 && threadIdx.x == 0) *alpha_j = r;
        }
        cf.gamma = SCALE ? (*gamma_num / *gamma_den) : 1.0;  // lbfgs.rs:691 ys/yy
        return cf;
    }
    __device__ void elem(const Coef& cf, const double* v, double* w, double* acc, uint64_t gi) const {
        const double src = NEG_SRC ? -v[0] : v[0];
        double q = src + cf.c * v[1];          // math.rs:35
        if constexpr (SCALE) q = q * cf.gamma; // math.rs:47
        const double wv = (VMODE == 0) ? v[2] : (VMODE == 1 ? v[1] : q);
        acc[0] += wv * q;                      // math.rs:41
        if constexpr (VMODE == 2) acc[1] += v[2] * q;  // g.d
        if constexpr (VMODE == 3) {
            if (gi >= owl_start && gi < owl_end && signum0(q) != signum0(-v[2])) q = 0.0;
            acc[2] += q * q;
            acc[3] += v[2] * q;                // pg.d (core.rs:90)
        }
        w[0] = q;
    }
};

// ---------------------------------------------------------------- vector-free (Gram) two-loop  [EXTENSION]
// The same recursion (lbfgs.rs:569-604) carried out in the coefficient space of the basis
// B = [s_0..s_{m-1}, y_0..y_{m-1}, g]: every dot product the recursion needs is an entry of B^T B, so ONE
// pass over the 2m+1 vectors refreshes the three Gram rows that changed (new s, new y, g), a scalar kernel
// runs the recursion on (2m+1) coefficients, and ONE more pass forms d = sum_j delta_j b_j.
// 4m+3 passes and 2 all-reduces instead of 8m+1 passes and 2m+1 all-reduces.  Mathematically identical,
// rounding differs (measured <= 1e-12 relative against the exact recursion; tests bound it by 1e-10).
// The rows pass is a dedicated kernel (gram_rows_kernel in lbfgs_hip.hip); the combine pass is an operator.
template <int M>
struct GramCoef {
    double c[2 * M + 1];
};
template <int M>
struct OpGramCombine {  // in[] in basis order; d = sum_j delta_j * b_j (ascending j); ||d||^2 ; g.d
    static constexpr int NB = 2 * M + 1;
    static constexpr int NIN = NB, NOUT = 1, NRED = 4;
    static constexpr int TUNE_MAP = 1, TUNE_UNROLL = 1;
    const double* in[NB];
    double* out[1];
    const double* delta;  // NB coefficients written by the scalar recursion kernel
    const double* pred;   // the coefficient-space prediction of ||d||^2: the third "sum" (see gram_combine.h GramCombArgs::pred)
    typedef GramCoef<M> Coef;
    __device__ Coef setup() const {
        Coef cf;
#pragma unroll
        for (int j = 0; j < NB; ++j) cf.c[j] = delta[j];
        return cf;
    }
    __device__ void elem(const Coef& cf, const double* v, double* w, double* acc, uint64_t gidx) const {
        double q = cf.c[0] * v[0];
#pragma unroll
        for (int j = 1; j < NB; ++j) q = q + cf.c[j] * v[j];
        w[0] = q;
        acc[0] += q * q;
        acc[1] += v[NB - 1] * q;
        if (gidx == 0) { acc[2] += pred[0]; acc[3] += pred[1]; }  // (element 0 of the global vector: one contribution in the whole job)
    }
};

// ---------------------------------------------------------------- OWL-QN (orthantwise.rs, core.rs)
// core.rs:123-126 after the user's evaluate: sum c*|x| (orthantwise.rs:70-79), pseudo-gradient
// (orthantwise.rs:82-112), ||pg||^2 (core.rs:185) and ||x||^2 (core.rs:193).  2r 1w.
struct OpOwlPost {
    static constexpr int NIN = 2, NOUT = 1, NRED = 3;
    const double* in[2];  // x, g
    double* out[1];       // pg
    double c;
    uint64_t start, end;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t gi) const {
        const double x = v[0], g = v[1];
        double pg = g;
        if (gi >= start && gi < end) {
            acc[0] += c * fabs(x);
            if (x != 0.0) {
                // f64::signum of a non-zero x (NaN propagates)
                const double sg = (x != x) ? x : ((__double_as_longlong(x) < 0) ? -1.0 : 1.0);
                pg = g + sg * c;
            } else {
                const double right_partial = g + c, left_partial = g - c;
                pg = (right_partial < 0.0) ? right_partial : ((left_partial > 0.0) ? left_partial : 0.0);
            }
        }
        w[0] = pg;
        acc[1] += pg * pg;
        acc[2] += x * x;
    }
};

// core.rs:167-180 over ALL i
struct OpOrthantSelect {
    static constexpr int NIN = 2, NOUT = 1, NRED = 0;
    const double* in[2];  // xp, pg
    double* out[1];       // wp
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double*, uint64_t) const {
        w[0] = (v[0] == 0.0) ? signum0(-v[1]) : signum0(v[0]);
    }
};

// orthantwise.rs:140-161: d_i = 0 where signum(d_i) != signum(-pg_i) on [start,end); ||d||^2 and pg.d
// (the next line search's dginit under OWL-QN, core.rs:90)
struct OpConstrainDir {
    static constexpr int NIN = 2, NOUT = 1, NRED = 2;
    const double* in[2];  // d, pg
    double* out[1];       // d
    uint64_t start, end;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t gi) const {
        double d = v[0];
        if (gi >= start && gi < end && signum0(d) != signum0(-v[1])) d = 0.0;
        w[0] = d;
        acc[0] += d * d;
        acc[1] += v[1] * d;
    }
};

// ---------------------------------------------------------------- device-resident objectives
// The synthetic workloads of BASELINE.json configs 2-4.  Data comes from a counter-based hash of
// the GLOBAL index, built from + and * only, so every rank and the CPU oracle (an independent C
// restatement in oracle/objectives.c) generate bit-identical a_i, b_i, t_i with nothing stored.
// splitmix64 of (seed, i): z = seed + (i + 1) * golden, two xor-shift-multiply rounds.  `hp` = (i + 1) * IDX_GOLDEN, the index
// product, comes from the skeleton (stream.h wants_index_product), which forms it once per thread and trip.
__device__ __forceinline__ uint64_t mix64_h(uint64_t seed, uint64_t hp) {
    uint64_t z = seed + hp;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static_assert(IDX_GOLDEN == 0x9E3779B97F4A7C15ULL, "the index multiplier of splitmix64 (oracle/objectives.c)");
// the top 53 bits of a hash as a double: v = m >> 11 < 2^53 converts EXACTLY as (2^84 + hi * 2^32) - (2^84 + 2^52) + (2^52 + lo)
// -- two words OR-ed under two exponents, two exact additions -- instead of two integer conversions, a scaling and an
// addition.  -> v itself (the caller folds the 2^-53 into its own multiplier: a power of two, so the rounding is the same)
__device__ __forceinline__ double hash_top53(uint64_t m) {
    const uint64_t v = m >> 11;
    const double hi = __longlong_as_double((long long)(0x4530000000000000ULL | (v >> 32)));
    const double lo = __longlong_as_double((long long)(0x4330000000000000ULL | (v & 0xffffffffULL)));
    return (hi - 0x1.00000001p+84) + lo;
}
constexpr double TWO_M53 = 1.0 / 9007199254740992.0;

struct ObjQuadratic {  // f_i = x*(0.5*a*x - b), g_i = a*x - b ; a = 1 + 999*u^2, b = 2*u' - 1
    uint64_t seed_a, seed_b;
    __device__ void eval(double x, uint64_t hp, double& f, double& g) const {
        const double ua = hash_top53(mix64_h(seed_a, hp)) * TWO_M53;
        const double a = 1.0 + 999.0 * (ua * ua);
        const double b = (2.0 * TWO_M53) * hash_top53(mix64_h(seed_b, hp)) - 1.0;   // = 2 u - 1: scaling by 2 is exact
        const double t = a * x;
        g = t - b;
        f = x * (0.5 * t - b);
    }
};

// softplus(-z) = log(1 + exp(-z)) and sigma(-z) = 1 / (1 + exp(z)) in f64, branch-free, from ONE range reduction -- the
// benchmark objective of BASELINE config 3 (the CALLER's closure kept on the device, not the reference's path; the oracle's
// restatement, oracle/objectives.c, calls glibc).  ocml's exp + log1p + an IEEE division cost ~170 issue slots per element and
// made every kernel that evaluates this objective ALU-bound at 2.4 TB/s (profiles/r05_probe_alu_check.log); this form takes
// ~60: within 4.2e-16 of the exact value (softplus) / 3.6e-16 (sigma) over |z| <= 745, mean 5e-17 / 6e-17 -- glibc's own
// sequence: 2.2e-16 / 3.3e-16 (tools/logistic_math_check.c emulates it bit for bit on the host; tests hold the device to it).
// (Both quotients are refined once against their exact residuals: without that -- 6.0e-16 / 4.8e-16, five operations fewer --
// two of 12 000 random logistic runs left the calibrated free-running bar of the test suite, profiles/r06_fuzz_soak*.log.)
//   e = exp(-|z|):      k = rint(t log2 e), r = t - k ln2 (two-part ln2: k ln2_hi exact), e = 2^k (1 + r + r^2 P9(r))
//   u = 1 + e in (1,2]: u > sqrt 2 ? (N, D, kk) = (e - 1, e + 3, 1) : (e, e + 2, 0), so that s = N / D = (m-1)/(m+1) with
//                       m = u / 2^kk in (0.707, 1.414]: log u = kk ln2 + 2 atanh s = kk ln2 + 2 s + s^3 Q6(s^2), |s| <= 0.1716
//   ONE reciprocal:     w = 1 / (u D) (v_rcp_f64 + one third-order step); 1/u = D w; s = N u w; each refined once (fma residual)
// NaN propagates through every step; z = +-inf, +-0 and |z| > 745 (e denormal or 0) give the oracle's values exactly.
#ifndef LH_LOGISTIC_REFINE
#define LH_LOGISTIC_REFINE 1  // 0: the quotients as they come (~2.5 ulp): 5 operations fewer, 6.0e-16 / 4.8e-16 (A/B builds)
#endif
struct LogisticMath {
    __device__ static __forceinline__ void eval(const double z, double& softplus, double& sigma) {
        double t = -__builtin_fabs(z);
        t = (t < -800.0) ? -800.0 : t;                        // exp underflows to 0 beyond -745.2 (a compare, not fmax: NaN stays)
        const double kf = __builtin_rint(t * 0x1.71547652b82fep+0);
        double r = __builtin_fma(-kf, 0x1.62e42fefa3000p-1, t);
        r = __builtin_fma(-kf, 0x1.3de6af278ece6p-42, r);     // |r| <= ln2 / 2
        double p = 0x1.af38a9b0ec855p-26;
        p = __builtin_fma(p, r, 0x1.289185613a3d6p-22);
        p = __builtin_fma(p, r, 0x1.71de0dae63bb3p-19);
        p = __builtin_fma(p, r, 0x1.a019b90d2ae7ap-16);
        p = __builtin_fma(p, r, 0x1.a01a01a7c41d5p-13);
        p = __builtin_fma(p, r, 0x1.6c16c1788bd90p-10);
        p = __builtin_fma(p, r, 0x1.11111111109b3p-7);
        p = __builtin_fma(p, r, 0x1.5555555553d63p-5);
        p = __builtin_fma(p, r, 0x1.5555555555556p-3);
        p = __builtin_fma(p, r, 0x1.0000000000001p-1);
        const double q = __builtin_fma(r * r, p, r);
        // k as an integer without a float-to-int conversion (of a NaN it would be undefined): kf + 1.5 * 2^52 carries k, two's
        // complement, in the low word of the sum (|k| <= 1155); a NaN stays a NaN through the ldexp whatever that word holds
        const int k = __double2loint(kf + 0x1.8p52);
        const double e = __builtin_amdgcn_ldexp(1.0 + q, k);
        const double u = 1.0 + e;
        const bool big = e > 0x1.a827999fcef32p-2;             // u > sqrt 2
        const double kk = big ? 1.0 : 0.0;
        const double N = e - kk;
        const double D = e + (big ? 3.0 : 2.0);
        const double pd = u * D;
        double w = __builtin_amdgcn_rcp(pd);                    // ~2^-23 (measured: tests/test_gpu_parity.py holds the result to 2e-15)
        const double e0 = __builtin_fma(-pd, w, 1.0);
        w = __builtin_fma(w, __builtin_fma(e0, e0, e0), w);     // one third-order step: w (1 + e0 + e0^2), error e0^3
        double inv_u = D * w;
        double s = (N * u) * w;
#if LH_LOGISTIC_REFINE
        // both quotients once more against their exact residuals (fma): 1/u and s = N/D to ~1 ulp instead of ~2.5
        inv_u = __builtin_fma(inv_u, __builtin_fma(-u, inv_u, 1.0), inv_u);
        s = __builtin_fma(__builtin_fma(-D, s, N), u * w, s);
#endif
        const double s2 = s * s;
        double l = 0x1.2b5f68a50d903p-3;
        l = __builtin_fma(l, s2, 0x1.39fdcceb4bb45p-3);
        l = __builtin_fma(l, s2, 0x1.7462b91b8df65p-3);
        l = __builtin_fma(l, s2, 0x1.c71c62d5e53e0p-3);
        l = __builtin_fma(l, s2, 0x1.2492492dfd86cp-2);
        l = __builtin_fma(l, s2, 0x1.9999999995273p-2);
        l = __builtin_fma(l, s2, 0x1.5555555555558p-1);
        double lg = __builtin_fma(kk, 0x1.abc9e3b39803fp-56, (s * s2) * l);
        lg = __builtin_fma(2.0, s, lg);
        lg = __builtin_fma(kk, 0x1.62e42fefa39efp-1, lg);
        softplus = lg + __builtin_fmax(-z, 0.0);                // z < 0: log(1 + e^-z) = -z + log(1 + e^z)
        sigma = ((z >= 0.0) ? e : 1.0) * inv_u;
    }
};

struct ObjLogistic {  // f_i = log(1+exp(-w*x)), w = +-(0.5 + 1.5*u)
    uint64_t seed_a, seed_b;
    static constexpr int GRID_X32 = 96;  // ~140 issue slots per element: 3 workgroups per CU hide them behind the loads
    __device__ void eval(double x, uint64_t hp, double& f, double& g) const {
        // a = 0.5 + 1.5 u with u = v 2^-53: (1.5 * 2^-53) v rounds like 1.5 (v 2^-53) -- the power of two is exact
        const double a = 0.5 + (1.5 * TWO_M53) * hash_top53(mix64_h(seed_a, hp));
        // w = bit 63 of the second hash ? a : -a: its complement goes straight into a's sign bit (a > 0)
        const unsigned int sgn = ~(unsigned int)(mix64_h(seed_b, hp) >> 32) & 0x80000000u;
        const double w = __hiloint2double(__double2hiint(a) | (int)sgn, __double2loint(a));
        const double z = w * x;
#if defined(LH_LOGISTIC_OCML)  // the libm form (A/B builds: tools/build_variants.sh "ocml=-DLH_LOGISTIC_OCML=1")
        const double e = exp(-fabs(z));
        double fi = log1p(e);
        double sig;
        if (z >= 0.0) {
            sig = e / (1.0 + e);
        } else {
            sig = 1.0 / (1.0 + e);
            fi -= z;
        }
#else
        double fi, sig;
        LogisticMath::eval(z, fi, sig);
#endif
        g = -w * sig;
        f = fi;
    }
};

// workgroups per 32 CUs for an operator that evaluates Obj per element: an objective with real arithmetic names its own
// (Obj::GRID_X32), the others take the operator's default
template <class Obj, int DEFAULT>
struct obj_grid {
    template <class T> static constexpr int get(decltype(T::GRID_X32)*) { return T::GRID_X32; }
    template <class T> static constexpr int get(...) { return DEFAULT; }
    static constexpr int value = get<Obj>(nullptr);
};

template <class Obj>
struct OpObjEval {  // core.rs:119-121 with the closure resident on the device.  1r 1w.
    static constexpr int NIN = 1, NOUT = 1, NRED = 1;
    static constexpr int TUNE_GRID_X32 = obj_grid<Obj, 27>::value;
    const double* in[1];  // x
    double* out[1];       // g
    Obj obj;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    static constexpr bool INDEX_PRODUCT = true;  // hp = (gi + 1) * IDX_GOLDEN from the skeleton (stream.h)
    __device__ void elem(const Coef& cf, const double* v, double* w, double* acc, uint64_t gi) const {
        elem(cf, v, w, acc, gi, (gi + 1) * IDX_GOLDEN);
    }
    __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t gi, uint64_t hp) const {
        double f, g;
        obj.eval(v[0], hp, f, g);
        w[0] = g;
        acc[0] += f;
    }
};

template <class Obj>
struct OpObjLineEval {  // take_line_step + evaluate + dg_unchecked (core.rs:155-158,119-121,114-116).  2r 2w.
    static constexpr int NIN = 2, NOUT = 2, NRED = 2;
    static constexpr int TUNE_GRID_X32 = obj_grid<Obj, 27>::value;
    const double* in[2];  // xp, d
    double* out[2];       // x, g
    double step;
    Obj obj;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    static constexpr bool INDEX_PRODUCT = true;  // hp = (gi + 1) * IDX_GOLDEN from the skeleton (stream.h)
    __device__ void elem(const Coef& cf, const double* v, double* w, double* acc, uint64_t gi) const {
        elem(cf, v, w, acc, gi, (gi + 1) * IDX_GOLDEN);
    }
    __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t gi, uint64_t hp) const {
        const double x = v[0] + step * v[1];
        double f, g;
        obj.eval(x, hp, f, g);
        w[0] = x;
        w[1] = g;
        acc[0] += f;
        acc[1] += g * v[1];
    }
};

// A trial step that leaves NO vectors behind: f(xp + step*d) and grad.d are all a line search reads from a
// trial (line.rs:283-288), so a search that needs T trials moves 2T passes instead of 4T.  The accepted point is
// materialised by OpHistUpdateFromStep below, which needs x and g anyway.  Same arithmetic as OpObjLineEval.  2r 0w.
template <class Obj>
struct OpObjLineProbe {
    static constexpr int NIN = 2, NOUT = 0, NRED = 2;
    static constexpr int TUNE_GRID_X32 = 96;  // ALU close behind HBM on the hashed objectives: 3 workgroups per CU
                                              // (profiles/r01_tune_objective_kernels.log)
    const double* in[2];  // xp, d
    double* out[1];
    double step;
    Obj obj;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    static constexpr bool INDEX_PRODUCT = true;  // hp = (gi + 1) * IDX_GOLDEN from the skeleton (stream.h)
    __device__ void elem(const Coef& cf, const double* v, double* w, double* acc, uint64_t gi) const {
        elem(cf, v, w, acc, gi, (gi + 1) * IDX_GOLDEN);
    }
    __device__ void elem(const Coef&, const double* v, double*, double* acc, uint64_t gi, uint64_t hp) const {
        const double x = v[0] + step * v[1];
        double f, g;
        obj.eval(x, hp, f, g);
        acc[0] += f;
        acc[1] += g * v[1];
    }
};

// take_line_step + evaluate at the ACCEPTED step (core.rs:155-158, :119-121) fused with IterationData::update
// (lbfgs.rs:640-673): x = xp + t*d ; g = grad f(x) ; s = x - xp ; y = g - gp and OpHistUpdate's seven sums.
// 3r 4w, where line eval (2r 2w) + update (4r 2w) move 6r 4w.  s is formed as (xp + t*d) - xp, never as t*d.
template <class Obj, bool DAMP>
struct OpHistUpdateFromStep {
    static constexpr int NIN = 3, NOUT = 4, NRED = 7;
    // default map / unroll, 2 workgroups per CU.  (The stand-alone sweep profiles/r01_tune_objective_kernels.log prefers
    // map 1 / unroll 2 / 1 per CU; inside bench.py that setting measured 1082 us against 925 us: not taken.)
    static constexpr int TUNE_GRID_X32 = obj_grid<Obj, 64>::value;
    const double* in[3];  // xp, d, gp
    double* out[4];       // x, g, s, y
    double t;             // the accepted trial step
    double neg_step;      // -step handed to update() (lbfgs.rs:671)
    Obj obj;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    static constexpr bool INDEX_PRODUCT = true;  // hp = (gi + 1) * IDX_GOLDEN from the skeleton (stream.h)
    __device__ void elem(const Coef& cf, const double* v, double* w, double* acc, uint64_t gi) const {
        elem(cf, v, w, acc, gi, (gi + 1) * IDX_GOLDEN);
    }
    __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t gi, uint64_t hp) const {
        const double x = v[0] + t * v[1];
        double f, g;
        obj.eval(x, hp, f, g);
        const double s = x - v[0];
        const double y = g - v[2];
        w[0] = x; w[1] = g; w[2] = s; w[3] = y;
        acc[0] += s * s;
        acc[1] += y * s;
        acc[2] += y * y;
        acc[3] += x * x;
        acc[4] += g * g;
        if constexpr (DAMP) acc[5] += s * (v[2] * neg_step);
        acc[6] += s * (-g);
    }
};

// OWL-QN trial in ONE pass (line.rs:740-743 with core.rs:155-164, :119-126, :114-116):
//   x = project(xp + step*d ; wp) ; g = grad f(x) ; pg = pseudo-gradient(x, g)
//   sums: f, g.d, c*sum|x|, ||pg||^2, ||x||^2                                  3r 3w (separately: 6r 3w)
// FIRST = true: the FIRST trial of a search also does Problem::update_orthant_new_point (line.rs:735, core.rs:167-180): the
//   orthant wp_i = xp_i == 0 ? signum(-pg_i) : signum(xp_i) is formed on the fly from xp (streamed anyway) and the PREVIOUS
//   pseudo-gradient -- in[2] is pg, read before this element's new pg is written over it -- and stored for the later trials.
//   3r 4w, where orthant_select (2r 1w) + the trial (3r 3w) move 5r 4w.
// UPD = true (round 6): the trial ALSO does IterationData::update for its point (lbfgs.rs:640-656), speculatively: s = x - xp,
//   y = g - gp into the history slot the update will fill, and ||s||^2, y.s, y.y.  A backtracking search accepts the trial it
//   evaluated last (line.rs:747-777), so the last trial's s, y and sums ARE the update's; under OWL-QN the first trial is
//   accepted almost always (config 3: 1.00 trials per search), and a rejected trial has cost one read and two writes more.
//   +1r 2w, where the update's own kernel moves 4r 2w.  (Not with Powell damping: its s.bs would be a ninth sum.)
template <class Obj, bool FIRST = false, bool UPD = false>
struct OpObjOwlLineEval {
    static constexpr int NIN = 3 + (UPD ? 1 : 0), NOUT = 3 + (FIRST ? 1 : 0) + (UPD ? 2 : 0), NRED = 5 + (UPD ? 3 : 0);
    static constexpr int TUNE_GRID_X32 = obj_grid<Obj, 27>::value;
    static constexpr int O_S = 3 + (FIRST ? 1 : 0);  // out[O_S] = s, out[O_S + 1] = y
    const double* in[4];  // xp, d, wp (FIRST: the previous pg), UPD: gp
    double* out[6];       // x, g, pg (FIRST: + wp) (UPD: + s, y)
    double step, c;
    uint64_t start, end;
    Obj obj;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    static constexpr bool INDEX_PRODUCT = true;  // hp = (gi + 1) * IDX_GOLDEN from the skeleton (stream.h)
    __device__ void elem(const Coef& cf, const double* v, double* w, double* acc, uint64_t gi) const {
        elem(cf, v, w, acc, gi, (gi + 1) * IDX_GOLDEN);
    }
    __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t gi, uint64_t hp) const {
        double x = v[0] + step * v[1];
        const bool reg = gi >= start && gi < end;
        double orthant = v[2];
        if constexpr (FIRST) {
            orthant = (v[0] == 0.0) ? signum0(-v[2]) : signum0(v[0]);  // core.rs:171-178, over ALL i
            w[3] = orthant;
        }
        if (reg && signum0(x) != signum0(orthant)) x = 0.0;       // orthantwise.rs:165-171
        double f, g;
        obj.eval(x, hp, f, g);
        double pg = g;
        if (reg) {                                             // orthantwise.rs:70-112
            acc[2] += c * fabs(x);
            if (x != 0.0) {
                const double sg = (x != x) ? x : ((__double_as_longlong(x) < 0) ? -1.0 : 1.0);
                pg = g + sg * c;
            } else {
                const double right_partial = g + c, left_partial = g - c;
                pg = (right_partial < 0.0) ? right_partial : ((left_partial > 0.0) ? left_partial : 0.0);
            }
        }
        w[0] = x; w[1] = g; w[2] = pg;
        acc[0] += f;
        acc[1] += g * v[1];
        acc[3] += pg * pg;
        acc[4] += x * x;
        if constexpr (UPD) {                                   // lbfgs.rs:642-654 (OpHistUpdate's arithmetic)
            const double s = x - v[0];
            const double y = g - v[3];
            w[O_S] = s; w[O_S + 1] = y;
            acc[5] += s * s;
            acc[6] += y * s;
            acc[7] += y * y;
        }
    }
};

// src/lib.rs:79-94 default_evaluate: couples (x[2i], x[2i+1]); n and the shard offset must be even.
struct RosenPair {
    __device__ static void eval(double x0, double x1, double& f, double& g0, double& g1) {
        const double t1 = 1.0 - x0;
        const double t2 = 10.0 * (x1 - x0 * x0);
        g1 = 20.0 * t2;
        g0 = -2.0 * (x0 * g1 + t1);
        f = t1 * t1 + t2 * t2;
    }
};
struct OpRosenEval {
    static constexpr bool PAIRWISE = true;
    static constexpr int NIN = 1, NOUT = 1, NRED = 1;
    const double* in[1];
    double* out[1];
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void pair(const Coef&, const d2* v, d2* w, double* acc, uint64_t) const {
        double f, g0, g1;
        RosenPair::eval(v[0].x, v[0].y, f, g0, g1);
        w[0].x = g0;
        w[0].y = g1;
        acc[0] += f;
    }
};
struct OpRosenLineEval {
    static constexpr bool PAIRWISE = true;
    static constexpr int NIN = 2, NOUT = 2, NRED = 2;
    const double* in[2];  // xp, d
    double* out[2];       // x, g
    double step;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void pair(const Coef&, const d2* v, d2* w, double* acc, uint64_t) const {
        const double x0 = v[0].x + step * v[1].x, x1 = v[0].y + step * v[1].y;
        double f, g0, g1;
        RosenPair::eval(x0, x1, f, g0, g1);
        w[0].x = x0;
        w[0].y = x1;
        w[1].x = g0;
        w[1].y = g1;
        acc[0] += f;
        acc[1] += g0 * v[1].x;
        acc[1] += g1 * v[1].y;
    }
};

// ---------------------------------------------------------------- small helpers used by the C-ABI
struct OpFill {  // vec![c; n]
    static constexpr int NIN = 0, NOUT = 1, NRED = 0;
    const double* in[1];
    double* out[1];
    double c;
    typedef NoCoef Coef;
    __device__ Coef setup() const { return {}; }
    __device__ void elem(const Coef&, const double*, double* w, double*, uint64_t) const { w[0] = c; }
};

struct OpScaleDevNorm {  // d = (-g)*gamma, ||d||^2 and g.d: two-loop with no corrections stored
    static constexpr int NIN = 1, NOUT = 1, NRED = 2;
    const double* in[1];  // g
    double* out[1];       // d
    const double *gn, *gd;
    typedef double Coef;
    __device__ Coef setup() const { return *gn / *gd; }
    __device__ void elem(const Coef& c, const double* v, double* w, double* acc, uint64_t) const {
        w[0] = (-v[0]) * c;
        acc[0] += w[0] * w[0];
        acc[1] += v[0] * w[0];
    }
};

// the reference's UNFUSED two-loop (lbfgs.rs:582-601 as written), coefficient formed on the device
struct OpAxpyAlpha {  // y += c*x with c formed on the device like the fused step does
    static constexpr int NIN = 2, NOUT = 1, NRED = 0;
    const double* in[2];
    double* out[1];
    const double* dot_in;
    const double* ys_j;
    double* alpha_j;
    int mode_b;
    typedef double Coef;
    __device__ Coef setup() const {
        const double r = *dot_in / *ys_j;
        if (mode_b) return *alpha_j - r;
        if (blockIdx.x == 0 && threadIdx.x == 0) *alpha_j = r;
        return -r;
    }
    __device__ void elem(const Coef& c, const double* v, double* w, double*, uint64_t) const { w[0] = v[0] + c * v[1]; }
};
struct OpScaleDev {
    static constexpr int NIN = 1, NOUT = 1, NRED = 0;
    const double* in[1];
    double* out[1];
    const double *gn, *gd;
    typedef double Coef;
    __device__ Coef setup() const { return *gn / *gd; }
    __device__ void elem(const Coef& c, const double* v, double* w, double*, uint64_t) const { w[0] = v[0] * c; }
};

}  // namespace lh
