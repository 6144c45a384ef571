// rust-lbfgs_amd/csrc/gram.h -- the rows pass of the vector-free (Gram) two-loop [EXTENSION, SURVEY 8f-2].
#pragma once
#include "stream.h"

namespace lh {

constexpr int GRAM_ROWS_UNROLL = 1;  // measured: 21 streams are already deep enough (tools/tune_stream.hip)

// ---- vector-free (Gram) two-loop: the three refreshed rows of B^T B in ONE pass over the 2m+1 vectors ----
// 63 running sums per lane (m = 10) do not fit the register file next to 21 in-flight 16-byte loads, so the
// four waves of a workgroup walk the SAME 64 pairs and split the COLUMNS: wave w owns columns
// [w*CPW, (w+1)*CPW) and keeps 3*CPW sums.  The three row vectors are re-loaded by every wave (L1/L2 hits).
// Position 0 = new s, 1 = new y, 2 = g, 3.. = the other vectors; out[r*NB + p] = b_r . b_p.
template <int M>
struct GramRowsArgs {
    const double* in[2 * M + 1];
};

template <int M, bool NT, int UNR>
__global__ __launch_bounds__(BLOCK) void gram_rows_kernel(const GramRowsArgs<M> a, const uint64_t n, const RedCtl red) {
    constexpr int NB = 2 * M + 1, CPW = (NB + WAVES - 1) / WAVES, NK = 3 * NB;
    static_assert(NK <= MAX_RED, "partials buffer overflow");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c_lo = wave * CPW;
    const unsigned int p2p_epoch0 = red.p2p.world > 1 ? red.ctr->p2p_epoch : 0u;  // read in the prologue (stream.h)
    double acc[3][CPW];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < CPW; ++c) acc[r][c] = 0.0;

    // this wave's column pointers, fetched ONCE (wave-uniform); columns past NB alias the last vector and
    // their sums are simply never published, so the loop body is branch-free
    const double* pr0 = a.in[0];
    const double* pr1 = a.in[1];
    const double* pr2 = a.in[2];
    const double* pc[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) pc[c] = a.in[(c_lo + c < NB) ? (c_lo + c) : (NB - 1)];

    const uint64_t n2 = n >> 1;
    const uint64_t nch = n2 / 64;  // full chunks of 64 pairs (1 KiB per stream); the ragged rest is below
    uint64_t t = blockIdx.x;
    for (; t + (UNR - 1) * (uint64_t)gridDim.x < nch; t += (uint64_t)UNR * gridDim.x) {
        d2 rs[UNR], ry[UNR], rg[UNR], v[UNR][CPW];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint64_t p = (t + (uint64_t)u * gridDim.x) * 64 + lane;
            rs[u] = ld16<NT>(pr0, p); ry[u] = ld16<NT>(pr1, p); rg[u] = ld16<NT>(pr2, p);
#pragma unroll
            for (int c = 0; c < CPW; ++c) v[u][c] = ld16<NT>(pc[c], p);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                acc[0][c] += rs[u].x * v[u][c].x; acc[0][c] += rs[u].y * v[u][c].y;
                acc[1][c] += ry[u].x * v[u][c].x; acc[1][c] += ry[u].y * v[u][c].y;
                acc[2][c] += rg[u].x * v[u][c].x; acc[2][c] += rg[u].y * v[u][c].y;
            }
    }
    for (; t < nch; t += gridDim.x) {
        const uint64_t p = t * 64 + lane;
        const d2 rs = ld16<NT>(pr0, p), ry = ld16<NT>(pr1, p), rg = ld16<NT>(pr2, p);
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const d2 v = ld16<NT>(pc[c], p);
            acc[0][c] += rs.x * v.x; acc[0][c] += rs.y * v.y;
            acc[1][c] += ry.x * v.x; acc[1][c] += ry.y * v.y;
            acc[2][c] += rg.x * v.x; acc[2][c] += rg.y * v.y;
        }
    }
    if (blockIdx.x == 0) {  // ragged rest: the last <64 pairs and, for odd n, the last element (scalar loads)
        for (uint64_t e = nch * 128 + lane; e < n; e += 64) {
            const double rs = pr0[e], ry = pr1[e], rg = pr2[e];
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const double v = pc[c][e];
                acc[0][c] += rs * v; acc[1][c] += ry * v; acc[2][c] += rg * v;
            }
        }
    }
    // wave-level sums -> one partial per (sum, workgroup); lane 0 of each wave publishes its own columns
    __shared__ unsigned int s_last;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const double w = wave_sum(acc[r][c]);
            if (lane == 0 && c_lo + c < NB) store_agent(red.partials + (size_t)(r * NB + c_lo + c) * MAX_GRID + blockIdx.x, w);
        }
    if (lane == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned int G = gridDim.x;
    if (threadIdx.x == 0) {
        const unsigned int tk = __hip_atomic_fetch_add(red.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk == G - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u) return;
    // last workgroup: wave w finishes sums w, w+4, ... in a fixed order (lane-strided partials, then the wave tree)
    __shared__ double s_vals[NK];
    for (int k = wave; k < NK; k += WAVES) {
        double s = 0.0;
        for (unsigned int b = lane; b < G; b += 64) s += load_agent(red.partials + (size_t)k * MAX_GRID + b);
        s = wave_sum(s);
        if (lane == 0) s_vals[k] = s;
    }
    __syncthreads();
    if (red.p2p.world > 1) {  // close the NK sums across ranks before the kernel ends
        __shared__ unsigned int s_bits[P2P_MAX_WORLD][MAX_RED][2];
        p2p_exchange(red.p2p, p2p_epoch0, s_vals, NK, s_bits);
        if (threadIdx.x == 0) red.ctr->p2p_epoch = next_epoch(p2p_epoch0);
    }
    for (int k = threadIdx.x; k < NK; k += BLOCK) red.out_contig[k] = s_vals[k];
    if (threadIdx.x == 0) __hip_atomic_store(red.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


}  // namespace lh
