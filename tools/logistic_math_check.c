/* tools/logistic_math_check.c -- host emulation, operation for operation (IEEE fma), of ops.h LogisticMath: softplus and sigmoid
 * of the hashed logistic objective in f64 from one range reduction and one reciprocal, against glibc's sequence (what the oracle
 * runs) and against long double.  v_rcp_f64 is emulated PESSIMISTICALLY (a float-precision reciprocal, off by another 3e-8).
 *   gcc -O2 -mfma -ffp-contract=off tools/logistic_math_check.c -o /tmp/logistic_math_check -lm && /tmp/logistic_math_check
 * Coefficients: tools/logistic_math_coefficients.py (Chebyshev interpolation in 60-digit arithmetic, rounded to double). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static const double LOG2E = 0x1.71547652b82fep+0, LN2_HI = 0x1.62e42fefa3000p-1, LN2_LO = 0x1.3de6af278ece6p-42;
static const double LN2F_HI = 0x1.62e42fefa39efp-1, LN2F_LO = 0x1.abc9e3b39803fp-56, SQRT2M1 = 0x1.a827999fcef32p-2;
static const double EP[10] = {0x1.0000000000001p-1, 0x1.5555555555556p-3, 0x1.5555555553d63p-5, 0x1.11111111109b3p-7, 0x1.6c16c1788bd90p-10, 0x1.a01a01a7c41d5p-13, 0x1.a019b90d2ae7ap-16, 0x1.71de0dae63bb3p-19, 0x1.289185613a3d6p-22, 0x1.af38a9b0ec855p-26};
static const double LQ[7] = {0x1.5555555555558p-1, 0x1.9999999995273p-2, 0x1.2492492dfd86cp-2, 0x1.c71c62d5e53e0p-3, 0x1.7462b91b8df65p-3, 0x1.39fdcceb4bb45p-3, 0x1.2b5f68a50d903p-3};
static double rcp_seed(double x) { return (double)(float)(1.0 / x) * (1.0 + 3e-8); }  /* pessimistic v_rcp_f64 */
static void fast(double z, double* sp, double* sig) {
    double t = -fabs(z);
    t = (t < -800.0) ? -800.0 : t;
    double kf = rint(t * LOG2E);
    double r = fma(-kf, LN2_HI, t);
    r = fma(-kf, LN2_LO, r);
    double p = EP[9];
    for (int i = 8; i >= 0; --i) p = fma(p, r, EP[i]);
    double q = fma(r * r, p, r);
    union { double d; int32_t w[2]; } kb; kb.d = kf + 0x1.8p52;   /* k from the low word of kf + 1.5 * 2^52 (no conversion of a NaN) */
    double e = ldexp(1.0 + q, kb.w[0]);
    double u = 1.0 + e;
    int big = e > SQRT2M1;
    double kk = big ? 1.0 : 0.0;
    double N = e - kk;
    double D = e + (big ? 3.0 : 2.0);
    double Pd = u * D;
    double w = rcp_seed(Pd);
    double e0 = fma(-Pd, w, 1.0);
    w = fma(w, fma(e0, e0, e0), w);
    double inv_u = D * w;
    double s = (N * u) * w;
#ifndef NO_REFINE                                             /* ops.h LH_LOGISTIC_REFINE (default 1) */
    inv_u = fma(inv_u, fma(-u, inv_u, 1.0), inv_u);          /* 1/u once more against its exact residual */
    s = fma(fma(-D, s, N), u * w, s);                         /* s += (N - D s) / D with 1/D ~ u w */
#endif
    double ww = s * s;
    double qq = LQ[6];
    for (int i = 5; i >= 0; --i) qq = fma(qq, ww, LQ[i]);
    double lg = fma(kk, LN2F_LO, (s * ww) * qq);
    lg = fma(2.0, s, lg);
    lg = fma(kk, LN2F_HI, lg);
    *sp = lg + fmax(-z, 0.0);
    *sig = ((z >= 0.0) ? e : 1.0) * inv_u;
    if (z != z) { *sp = z; *sig = z; }
}
static void glibc(double z, double* sp, double* sig) {
    double e = exp(-fabs(z));
    double fi = log1p(e);
    double s;
    if (z >= 0.0) s = e / (1.0 + e); else { s = 1.0 / (1.0 + e); fi -= z; }
    *sp = fi; *sig = s;
}
static void exact(double z, long double* sp, long double* sig) {
    long double e = expl(-fabsl((long double)z));
    long double fi = log1pl(e);
    *sig = (z >= 0) ? e / (1 + e) : 1 / (1 + e);
    *sp = (z < 0) ? fi - z : fi;
}
int main() {
    uint64_t st = 88172645463325252ull;
    double mf = 0, mg = 0, mf_g = 0, mg_g = 0, sumf = 0, sumg = 0; double wz_f = 0, wz_g = 0;
    long N = 20000000;
    for (long i = 0; i < N; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double u01 = (st >> 11) * (1.0 / 9007199254740992.0);
        double z;
        int cls = i % 4;
        if (cls == 0) z = (u01 * 2 - 1) * 4;            /* the workload's range */
        else if (cls == 1) z = (u01 * 2 - 1) * 40;
        else if (cls == 2) z = (u01 * 2 - 1) * 745;
        else z = (u01 * 2 - 1) * 1e-3;
        double a, b, c, d; long double ea, eb;
        fast(z, &a, &b); glibc(z, &c, &d); exact(z, &ea, &eb);
        if (ea > 1e-300L) { double r = fabsl((a - ea) / ea); if (r > mf) { mf = r; wz_f = z; } sumf += r; r = fabsl((c - ea) / ea); if (r > mf_g) mf_g = r; }
        if (eb > 1e-300L) { double r = fabsl((b - eb) / eb); if (r > mg) { mg = r; wz_g = z; } sumg += r; r = fabsl((d - eb) / eb); if (r > mg_g) mg_g = r; }
    }
    printf("fast:  max rel f %.3e (z=%g)  sig %.3e (z=%g); mean f %.3e sig %.3e\n", mf, wz_f, mg, wz_g, sumf / N, sumg / N);
    printf("glibc: max rel f %.3e  sig %.3e\n", mf_g, mg_g);
    double zs[] = {0.0, -0.0, 1e-320, -1e-320, 800, -800, 1e300, -1e300, INFINITY, -INFINITY, NAN, 0.8813735870195429, -0.8813735870195429, 37, -37, 708.5, -745.2};
    for (unsigned i = 0; i < sizeof zs / sizeof *zs; ++i) {
        double a, b, c, d; fast(zs[i], &a, &b); glibc(zs[i], &c, &d);
        printf("z=%-12g fast f=%-24.17g sig=%-24.17g | glibc f=%-24.17g sig=%-24.17g\n", zs[i], a, b, c, d);
    }
    return 0;
}
