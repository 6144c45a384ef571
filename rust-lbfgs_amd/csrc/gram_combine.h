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
    const double* pred;             // [2]: ||d||^2 as the coefficient-space arithmetic predicts it, its cancellation figure ...
    int lead;                       // ... which the thread that owns element 0 of the GLOBAL vector carries into two more "sums"
                                    // (every other contribution is 0: the total is the prediction, bit for bit), so that it
                                    // reaches the board -- and the host mirror -- next to the ||d||^2 summed here
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
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
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
    if (a.lead && B == 0 && tid == 0) { acc[2] = a.pred[0]; acc[3] = a.pred[1]; }
    grid_reduce<4>(acc, red, c0);
}


// ---- the rows pass of the vector-free two-loop, the same way ---------------------------------------------------------
// out[r * NB + p] = b_r . b_p for the three refreshed rows r (positions 0, 1, 2 = new s, new y, g) and all NB = 2m+1
// positions p.  gram.h's kernel reads the 2m+1 vectors of an element at once (21 concurrent streams: 5.8 TB/s); here a
// tile of the THREE ROW VECTORS (20 pairs per thread and row = 60 pairs = all the accumulation registers) stays on the
// chip while the other columns are streamed past it one after the other.  Every thread keeps its 3 * NB running sums in
// an LDS column of its own (129 KB per workgroup at m = 10); they meet in the usual two-stage reduction at the end.
constexpr int GR_T = 20;                   // pairs per thread, row and tile
constexpr int GR_TG = GR_T / RES_UNROLL;   // groups of four
static_assert(3 * GR_T * 4 <= 256 && GR_T % RES_UNROLL == 0, "three row tiles in the AGPRs");

struct GramRowsResArgs {
    const double* in[GC_MAX_COLS];  // position order: 0, 1, 2 = the rows
    uint64_t n;
    int nb;
    uint32_t total_rounds;          // ceil(pairs / (grid * BLOCK))
};

template <bool NT>
__device__ __forceinline__ void gr_fetch(d2 (&uu)[RES_UNROLL], const uint32_t group, const uint32_t p_first, const uint32_t p_stride,
                                         const uint32_t n2, const double* up) {
    const uint32_t p0 = res_opaque(p_first) + group * RES_UNROLL * p_stride;
#pragma unroll
    for (int u = 0; u < RES_UNROLL; ++u) {
        const uint32_t p = p0 + (uint32_t)u * p_stride;
        const bool in = p < n2;
        uu[u] = ld16<NT>(up, in ? p : 0u);  // (a valid address: masked below -- pairs past the end contribute zeros)
        if (!in) uu[u] = d2{0.0, 0.0};
    }
}

template <int G, bool NT>
struct GrGroups {
    // the row vectors' tile -> a[(R * GR_T + 4G + u) * 4 ..]
    static __device__ __forceinline__ void load_rows(const uint32_t p_first, const uint32_t p_stride, const uint32_t n2,
                                                     const double* r0, const double* r1, const double* r2) {
        if constexpr (G < GR_TG) {
            d2 a0[RES_UNROLL], a1[RES_UNROLL], a2[RES_UNROLL];
            gr_fetch<NT>(a0, G, p_first, p_stride, n2, r0);
            gr_fetch<NT>(a1, G, p_first, p_stride, n2, r1);
            gr_fetch<NT>(a2, G, p_first, p_stride, n2, r2);
            asm volatile("" ::: "memory");
            acc_put<0 * GR_T + G * RES_UNROLL + 0>(a0[0]); acc_put<0 * GR_T + G * RES_UNROLL + 1>(a0[1]);
            acc_put<0 * GR_T + G * RES_UNROLL + 2>(a0[2]); acc_put<0 * GR_T + G * RES_UNROLL + 3>(a0[3]);
            acc_put<1 * GR_T + G * RES_UNROLL + 0>(a1[0]); acc_put<1 * GR_T + G * RES_UNROLL + 1>(a1[1]);
            acc_put<1 * GR_T + G * RES_UNROLL + 2>(a1[2]); acc_put<1 * GR_T + G * RES_UNROLL + 3>(a1[3]);
            acc_put<2 * GR_T + G * RES_UNROLL + 0>(a2[0]); acc_put<2 * GR_T + G * RES_UNROLL + 1>(a2[1]);
            acc_put<2 * GR_T + G * RES_UNROLL + 2>(a2[2]); acc_put<2 * GR_T + G * RES_UNROLL + 3>(a2[3]);
            GrGroups<G + 1, NT>::load_rows(p_first, p_stride, n2, r0, r1, r2);
        }
    }
    // sum[r] += row_r . v over group G's four pairs (x then y of every pair, as gram.h)
    static __device__ __forceinline__ void dot3(const d2 (&v)[RES_UNROLL], double* sum) {
        const d2 s[RES_UNROLL] = {acc_get<0 * GR_T + G * RES_UNROLL + 0>(), acc_get<0 * GR_T + G * RES_UNROLL + 1>(),
                                  acc_get<0 * GR_T + G * RES_UNROLL + 2>(), acc_get<0 * GR_T + G * RES_UNROLL + 3>()};
        const d2 y[RES_UNROLL] = {acc_get<1 * GR_T + G * RES_UNROLL + 0>(), acc_get<1 * GR_T + G * RES_UNROLL + 1>(),
                                  acc_get<1 * GR_T + G * RES_UNROLL + 2>(), acc_get<1 * GR_T + G * RES_UNROLL + 3>()};
        const d2 g[RES_UNROLL] = {acc_get<2 * GR_T + G * RES_UNROLL + 0>(), acc_get<2 * GR_T + G * RES_UNROLL + 1>(),
                                  acc_get<2 * GR_T + G * RES_UNROLL + 2>(), acc_get<2 * GR_T + G * RES_UNROLL + 3>()};
#pragma unroll
        for (int u = 0; u < RES_UNROLL; ++u) {
            sum[0] += s[u].x * v[u].x; sum[0] += s[u].y * v[u].y;
            sum[1] += y[u].x * v[u].x; sum[1] += y[u].y * v[u].y;
            sum[2] += g[u].x * v[u].x; sum[2] += g[u].y * v[u].y;
        }
        res_pin(sum[0]); res_pin(sum[1]); res_pin(sum[2]);
    }
    // one column past the tile: `cur`, `nxt` = its loaded groups G, G+1 (two groups of loads ahead, as the combine pass)
    static __device__ __forceinline__ void col(d2 (&cur)[RES_UNROLL], d2 (&nxt)[RES_UNROLL], const uint32_t p_first,
                                               const uint32_t p_stride, const uint32_t n2, const double* up, double* sum) {
        if constexpr (G < GR_TG) {
            d2 far[RES_UNROLL];
            gr_fetch<NT>(far, (G + 2 < GR_TG) ? (uint32_t)(G + 2) : (uint32_t)G, p_first, p_stride, n2, up);  // (past the end: unused)
            asm volatile("" ::: "memory");
            dot3(cur, sum);
            GrGroups<G + 1, NT>::col(nxt, far, p_first, p_stride, n2, up, sum);
        }
    }
    // the rows among themselves (positions 0, 1, 2), from the chip: s9[r * 3 + p]
    static __device__ __forceinline__ void rows_rows(double* s9) {
        if constexpr (G < GR_TG) {
            const d2 s[RES_UNROLL] = {acc_get<0 * GR_T + G * RES_UNROLL + 0>(), acc_get<0 * GR_T + G * RES_UNROLL + 1>(),
                                      acc_get<0 * GR_T + G * RES_UNROLL + 2>(), acc_get<0 * GR_T + G * RES_UNROLL + 3>()};
            const d2 y[RES_UNROLL] = {acc_get<1 * GR_T + G * RES_UNROLL + 0>(), acc_get<1 * GR_T + G * RES_UNROLL + 1>(),
                                      acc_get<1 * GR_T + G * RES_UNROLL + 2>(), acc_get<1 * GR_T + G * RES_UNROLL + 3>()};
            const d2 g[RES_UNROLL] = {acc_get<2 * GR_T + G * RES_UNROLL + 0>(), acc_get<2 * GR_T + G * RES_UNROLL + 1>(),
                                      acc_get<2 * GR_T + G * RES_UNROLL + 2>(), acc_get<2 * GR_T + G * RES_UNROLL + 3>()};
#pragma unroll
            for (int u = 0; u < RES_UNROLL; ++u) {
                const d2 r[3] = {s[u], y[u], g[u]};
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) { s9[a * 3 + b] += r[a].x * r[b].x; s9[a * 3 + b] += r[a].y * r[b].y; }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) res_pin(s9[k]);
            GrGroups<G + 1, NT>::rows_rows(s9);
        }
    }
};

template <bool NT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gram_rows_resident_kernel(const GramRowsResArgs a, const RedCtl red) {
    asm volatile("" ::: "a255");
    extern __shared__ double gr_acc[];  // [3 * nb][BLOCK]: every thread's running sums
    const unsigned int p2p_epoch0 = red.p2p.world > 1 ? red.ctr->p2p_epoch : 0u;  // read in the prologue (stream.h)
    const int NB = a.nb, NK = 3 * NB;
    const uint32_t n2 = (uint32_t)(a.n >> 1);
    const uint32_t G = gridDim.x, B = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p_stride = G * BLOCK;
    for (int k = 0; k < NK; ++k) gr_acc[(size_t)k * BLOCK + tid] = 0.0;
    for (uint32_t e_lo = 0; e_lo < a.total_rounds; e_lo += GR_T) {
        const uint32_t p_first = (e_lo * G + B) * BLOCK + tid;
        GrGroups<0, NT>::load_rows(p_first, p_stride, n2, a.in[0], a.in[1], a.in[2]);
        {
            double s9[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            GrGroups<0, NT>::rows_rows(s9);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int p = 0; p < 3; ++p) gr_acc[(size_t)(r * NB + p) * BLOCK + tid] += s9[r * 3 + p];
        }
        for (int p = 3; p < NB; ++p) {
            const double* up = a.in[p];
            double sum[3] = {0.0, 0.0, 0.0};
            d2 cur[RES_UNROLL], nxt[RES_UNROLL];
            gr_fetch<NT>(cur, 0u, p_first, p_stride, n2, up);
            gr_fetch<NT>(nxt, 1u, p_first, p_stride, n2, up);
            GrGroups<0, NT>::col(cur, nxt, p_first, p_stride, n2, up, sum);
#pragma unroll
            for (int r = 0; r < 3; ++r) gr_acc[(size_t)(r * NB + p) * BLOCK + tid] += sum[r];
        }
    }
    if ((a.n & 1) && B == G - 1 && tid == 0) {  // odd n: the last element
        for (int r = 0; r < 3; ++r)
            for (int p = 0; p < NB; ++p) gr_acc[(size_t)(r * NB + p) * BLOCK + tid] += a.in[r][a.n - 1] * a.in[p][a.n - 1];
    }
    // ---- per-thread sums -> one partial per (sum, workgroup): wave tree, waves in order; then gram.h's hand-off
    __shared__ double s_w[WAVES][MAX_RED];
    __shared__ unsigned int s_last;
    for (int k = 0; k < NK; ++k) {
        const double w = wave_sum(gr_acc[(size_t)k * BLOCK + tid]);
        if (lane == 0) s_w[wave][k] = w;
    }
    __syncthreads();
    if ((int)tid < NK) {
        double t = s_w[0][tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t += s_w[w][tid];
        store_agent(red.partials + (size_t)tid * MAX_GRID + B, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned int tk = __hip_atomic_fetch_add(red.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk == G - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u) return;
    __shared__ double s_vals[MAX_RED];
    for (int k = wave; k < NK; k += WAVES) {  // (the last workgroup, as gram.h: lane-strided partials, then the wave tree)
        double s = 0.0;
        for (unsigned int b = lane; b < G; b += 64) s += load_agent(red.partials + (size_t)k * MAX_GRID + b);
        s = wave_sum(s);
        if (lane == 0) s_vals[k] = s;
    }
    __syncthreads();
    if (red.p2p.world > 1) {
        __shared__ unsigned int s_bits[P2P_MAX_WORLD][MAX_RED][2];
        p2p_exchange(red.p2p, p2p_epoch0, s_vals, NK, s_bits);
        if (tid == 0) red.ctr->p2p_epoch = next_epoch(p2p_epoch0);
    }
    for (int k = tid; k < NK; k += BLOCK) red.out_contig[k] = s_vals[k];
    if (tid == 0) __hip_atomic_store(red.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace lh
