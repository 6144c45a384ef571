// rust-lbfgs_amd/csrc/gram_combine.h -- the combine pass of the vector-free (Gram) two-loop [EXTENSION, SURVEY 8f-2],
//     d = sum_j delta_j * b_j (ascending j) ; ||d||^2 ; g.d         (b_{2m} = g is the last column)
// with d accumulated ON THE CHIP, tile by tile, in the accumulation registers + LDS that resident.h uses for the running
// vector of the exact recursion.  ops.h's OpGramCombine loads all 2m+1 columns of an element at once -- 21 concurrent
// DRAM streams, 5.3 TB/s at n = 1e8; here a tile of d (up to 96 pairs per thread, 100 MB over the chip) stays put while
// the columns are streamed through it ONE AFTER THE OTHER, the access pattern at which the persistent two-loop kernel
// reads 6.7 TB/s.  No synchronisation between workgroups until the usual final reduction: nothing crosses threads.
// Arithmetic per element as in OpGramCombine: q = 0 + c0*b0 ; q = q + c_j*b_j (a multiply, then an add); the two sums
// are formed in another order than the streaming kernel's (the extension's parity is a tolerance, not bit equality).
#pragma once
#include "resident.h"

namespace lh {

constexpr int GC_MAX_COLS = 2 * 24 + 1;
#ifndef LH_GC_AHEAD
#define LH_GC_AHEAD 2
#endif
constexpr int GC_AHEAD = LH_GC_AHEAD;  // groups whose loads are in flight ahead of the one being worked on (2 or 3)
constexpr int GC_ER = 60;  // pairs per thread in AGPRs (the same budget as resident.h)

struct GramCombArgs {
    const double* in[GC_MAX_COLS];  // basis order; the last one is g
    double* d;
    const double* delta;            // nb coefficients (device: written by the scalar recursion kernel)
    uint64_t n;
    int nb;
    uint32_t total_rounds;          // ceil(pairs / (grid * BLOCK))
    uint32_t tile_rounds;           // rounds per tile (<= GC_ER + RES_LDS_PAIRS_MAX)
};

// register groups of a tile, unrolled (the AGPR numbers are compile-time); groups at or beyond `er` are skipped (uniform)
template <int G, bool NT>
struct GcGroups {
    static constexpr int NG = GC_ER / RES_UNROLL;
    static __device__ __forceinline__ void zero(const uint32_t er) {
        if constexpr (G < NG) {
            if ((uint32_t)(G * RES_UNROLL) < er) {
                acc_put<G * RES_UNROLL + 0>(d2{0.0, 0.0}); acc_put<G * RES_UNROLL + 1>(d2{0.0, 0.0});
                acc_put<G * RES_UNROLL + 2>(d2{0.0, 0.0}); acc_put<G * RES_UNROLL + 3>(d2{0.0, 0.0});
                GcGroups<G + 1, NT>::zero(er);
            }
        }
    }
    static __device__ __forceinline__ void fetch(d2 (&uu)[RES_UNROLL], const uint32_t group, const uint32_t b_first,
                                                 const uint32_t b_stride, const double* up) {
        const uint32_t o0 = res_opaque(b_first) + group * RES_UNROLL * b_stride;
#pragma unroll
        for (int u = 0; u < RES_UNROLL; ++u) uu[u] = ld16_at<NT>(up, o0 + (uint32_t)u * b_stride);
    }
    // q += c*u over the tile's register rounds; LAST: also acc0 += q.q ; acc1 += u.q and d = q
    // (`cur`, `nxt`, `nx2`: the loaded groups G, G+1, G+2 -- the loads of group G+GC_AHEAD are issued before G is worked on:
    // with ONE column in flight at a time it takes 12-16 loads of 16 bytes per thread to keep HBM busy)
    template <bool LAST>
    static __device__ __forceinline__ void axpy(d2 (&cur)[RES_UNROLL], d2 (&nxt)[RES_UNROLL], d2 (&nx2)[RES_UNROLL],
                                                const uint32_t er, const uint32_t b_first, const uint32_t b_stride,
                                                const double* up, const double c, double* acc, double* dp) {
        if constexpr (G < NG) {
            if ((uint32_t)(G * RES_UNROLL) < er) {
                d2 far[RES_UNROLL];  // (past the end: this group again, unused)
                fetch(far, ((uint32_t)((G + GC_AHEAD) * RES_UNROLL) < er) ? (uint32_t)(G + GC_AHEAD) : (uint32_t)G, b_first, b_stride, up);
                asm volatile("" ::: "memory");
                d2 q[RES_UNROLL] = {acc_get<G * RES_UNROLL + 0>(), acc_get<G * RES_UNROLL + 1>(), acc_get<G * RES_UNROLL + 2>(),
                                    acc_get<G * RES_UNROLL + 3>()};
#pragma unroll
                for (int u = 0; u < RES_UNROLL; ++u) {
                    q[u].x = q[u].x + c * cur[u].x;
                    q[u].y = q[u].y + c * cur[u].y;
                    if constexpr (LAST) {
                        acc[0] += q[u].x * q[u].x; acc[1] += cur[u].x * q[u].x;
                        acc[0] += q[u].y * q[u].y; acc[1] += cur[u].y * q[u].y;
                        st16_at<NT>(dp, res_opaque(b_first) + (uint32_t)(G * RES_UNROLL + u) * b_stride, q[u]);
                    }
                }
                if constexpr (LAST) { res_pin(acc[0]); res_pin(acc[1]); }
                else {
                    acc_put<G * RES_UNROLL + 0>(q[0]); acc_put<G * RES_UNROLL + 1>(q[1]);
                    acc_put<G * RES_UNROLL + 2>(q[2]); acc_put<G * RES_UNROLL + 3>(q[3]);
                }
                if constexpr (GC_AHEAD == 3) GcGroups<G + 1, NT>::template axpy<LAST>(nxt, nx2, far, er, b_first, b_stride, up, c, acc, dp);
                else GcGroups<G + 1, NT>::template axpy<LAST>(nxt, far, far, er, b_first, b_stride, up, c, acc, dp);
            }
        }
    }
};

template <bool NT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gram_combine_resident_kernel(const GramCombArgs a, const RedCtl red) {
    asm volatile("" ::: "a255");  // (as resident.h: the wave owns the accumulation registers)
    extern __shared__ d2 gc_lds[];  // [RES_LDS_PAIRS_MAX][BLOCK]
    const DevCounters c0 = load_counters(red);
    const uint32_t n2 = (uint32_t)(a.n >> 1);
    const uint32_t G = gridDim.x, B = blockIdx.x, tid = threadIdx.x;
    const uint32_t p_stride = G * BLOCK;
    double acc[2] = {0.0, 0.0};
    for (uint32_t e_lo = 0; e_lo < a.total_rounds; e_lo += a.tile_rounds) {
        const uint32_t e_hi = min(e_lo + a.tile_rounds, a.total_rounds), rounds = e_hi - e_lo;
        // every round but the vector's last one is full: the register rounds carry no bounds checks, the rest go to LDS
        const uint32_t full = rounds - (e_hi == a.total_rounds ? 1u : 0u);
        const uint32_t er = min((uint32_t)GC_ER, full) & ~(uint32_t)(RES_UNROLL - 1), el = rounds - er;  // (el <= 36: host)
        const uint32_t p_first = (e_lo * G + B) * BLOCK + tid;       // first pair of this thread in this tile
        const uint32_t b_first = p_first * 16u, b_stride = p_stride * 16u;
        GcGroups<0, NT>::zero(er);
        for (uint32_t e = 0; e < el; ++e) gc_lds[(size_t)e * BLOCK + tid] = d2{0.0, 0.0};
        for (int j = 0; j < a.nb; ++j) {
            const double c = a.delta[j];
            const double* up = a.in[j];
            const bool last = j == a.nb - 1;
            if (er > 0) {
                d2 cur[RES_UNROLL], nxt[RES_UNROLL], nx2[RES_UNROLL];
                GcGroups<0, NT>::fetch(cur, 0u, b_first, b_stride, up);
                GcGroups<0, NT>::fetch(nxt, er > (uint32_t)RES_UNROLL ? 1u : 0u, b_first, b_stride, up);
                if constexpr (GC_AHEAD == 3) GcGroups<0, NT>::fetch(nx2, er > 2u * RES_UNROLL ? 2u : 0u, b_first, b_stride, up);
                if (last) GcGroups<0, NT>::template axpy<true>(cur, nxt, nx2, er, b_first, b_stride, up, c, acc, a.d);
                else GcGroups<0, NT>::template axpy<false>(cur, nxt, nx2, er, b_first, b_stride, up, c, acc, a.d);
            }
            // LDS rounds (the ragged last round among them: bounds-checked), two groups of loads ahead as well
            auto lfetch = [&](d2 (&uu)[RES_UNROLL], const uint32_t e0) {
#pragma unroll
                for (int u = 0; u < RES_UNROLL; ++u) {
                    const uint32_t p = p_first + (er + e0 + u) * p_stride;
                    uu[u] = ld16<NT>(up, ((e0 + u < el) && p < n2) ? p : 0u);  // (a valid address; masked where it is used)
                }
            };
            d2 l0[RES_UNROLL], l1[RES_UNROLL];
            if (el > 0) { lfetch(l0, 0u); lfetch(l1, RES_UNROLL); }
            for (uint32_t e0 = 0; e0 < el; e0 += RES_UNROLL) {
                d2 l2[RES_UNROLL];
                lfetch(l2, e0 + 2u * RES_UNROLL);  // (past the end: pair 0, unused)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < RES_UNROLL; ++u) {
                    const uint32_t p = p_first + (er + e0 + u) * p_stride;
                    if ((e0 + u < el) && p < n2) {
                        d2 q = gc_lds[(size_t)(e0 + u) * BLOCK + tid];
                        q.x = q.x + c * l0[u].x;
                        q.y = q.y + c * l0[u].y;
                        if (last) {
                            acc[0] += q.x * q.x; acc[1] += l0[u].x * q.x;
                            acc[0] += q.y * q.y; acc[1] += l0[u].y * q.y;
                            st16<NT>(a.d, p, q);
                        } else {
                            gc_lds[(size_t)(e0 + u) * BLOCK + tid] = q;
                        }
                    }
                }
                res_pin(acc[0]); res_pin(acc[1]);
#pragma unroll
                for (int u = 0; u < RES_UNROLL; ++u) { l0[u] = l1[u]; l1[u] = l2[u]; }
            }
        }
    }
    if ((a.n & 1) && B == G - 1 && tid == 0) {  // odd n: the last element
        double q = 0.0;
        for (int j = 0; j < a.nb; ++j) q = q + a.delta[j] * a.in[j][a.n - 1];
        const double gl = a.in[a.nb - 1][a.n - 1];
        acc[0] += q * q;
        acc[1] += gl * q;
        a.d[a.n - 1] = q;
    }
    grid_reduce<2>(acc, red, c0);
}

}  // namespace lh
