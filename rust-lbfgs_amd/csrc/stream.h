// rust-lbfgs_amd/csrc/stream.h -- the one streaming skeleton every hot-path kernel is built on.
//
// All L-BFGS / OWL-QN vector work is BLAS-1: each element is touched O(1) times,
// arithmetic intensity ~0.1-0.25 flop/byte, so the bound is HBM bandwidth and
// nothing here uses MFMA.  What matters on MI355X (gfx950):
//   * 16-byte-per-lane coalesced loads/stores (double2 -> global_load_dwordx4):
//     one wave instruction moves 1 KiB;
//   * enough bytes in flight per CU: every thread issues UNROLL independent
//     16-B loads per input stream before the first use (compiler places the
//     s_waitcnt at first use);
//   * a persistent grid of (CUs x blocks-per-CU) workgroups, each owning one
//     contiguous, evenly sized range of the vector (no tail imbalance, and the
//     block -> address-range map is identical in every launch);
//   * reductions: f64 per-thread accumulators -> wave64 shuffle tree -> LDS ->
//     one partial per workgroup -> the LAST workgroup to finish (agent-scope
//     ticket) sums the partials in a FIXED order.  No float atomics, so results
//     are bitwise reproducible for a given (n, grid).
//
// An operator (ops.h) only states: its input/output streams, how one element is
// computed, and how many sums it accumulates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lh {

constexpr int BLOCK = 256;          // 4 wave64 per workgroup
constexpr int WAVES = BLOCK / 64;
constexpr int UNROLL = 4;           // independent 16-B loads per stream per thread
constexpr int MAX_RED = 6;          // sums a kernel may produce
constexpr int MAX_GRID = 4096;      // upper bound on workgroups per launch

typedef double d2 __attribute__((ext_vector_type(2)));

struct RedCtl {
    double* partials;        // [MAX_RED][MAX_GRID] workgroup partial sums
    unsigned int* ticket;    // arrival counter, self-resetting
    double* out[MAX_RED];    // where the last workgroup puts the totals
};

// ---- agent-scope accesses for the cross-workgroup hand-off -------------------------------
// Partials are stored write-through (sc1) and read back with sc1 loads, the ticket is an
// agent-scope atomic: the "8-byte agent atomics on both sides" hand-off form, which needs no
// L2 write-back fence.  Every storing lane drains its stores (s_waitcnt vmcnt(0)) before the
// workgroup barrier that precedes the ticket add.
__device__ __forceinline__ void store_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p) {
    return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

// Sum `v` over the workgroup in a fixed order; result valid in thread 0.
template <int NRED>
__device__ __forceinline__ void block_sum(double (&acc)[NRED], double (*lds)[WAVES]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
        double w = wave_sum(acc[k]);
        if (lane == 0) lds[k][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) {
            double t = lds[k][0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) t += lds[k][w];
            acc[k] = t;
        }
    }
}

template <int NRED>
__device__ __forceinline__ void grid_reduce(double (&acc)[NRED], const RedCtl& red) {
    __shared__ double lds[NRED][WAVES];
    __shared__ unsigned int s_last;
    block_sum<NRED>(acc, lds);
    const unsigned int G = gridDim.x;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) store_agent(red.partials + (size_t)k * MAX_GRID + blockIdx.x, acc[k]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // partials have left this CU
        unsigned int t = __hip_atomic_fetch_add(red.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == G - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u) return;  // uniform per workgroup

    // last workgroup: every other workgroup's partials were drained before its ticket add
    double tot[NRED];
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
        double t = 0.0;
        for (unsigned int b = threadIdx.x; b < G; b += BLOCK) t += load_agent(red.partials + (size_t)k * MAX_GRID + b);
        tot[k] = t;
    }
    __syncthreads();  // lds reuse
    block_sum<NRED>(tot, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) *red.out[k] = tot[k];
        __hip_atomic_store(red.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the skeleton ----------------------------------------------------------------------
// Op interface:
//   static constexpr int NIN, NOUT, NRED;
//   const double* in[NIN]; double* out[NOUT];
//   typename Op::Coef; __device__ Coef setup() const;            // wave-uniform scalars (board reads)
//   __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t gidx) const;
//     v[NIN] inputs of one element, w[NOUT] outputs, acc[NRED] running sums, gidx = GLOBAL index
// An Op with `static constexpr bool PAIRWISE = true` instead implements
//   __device__ void pair(const Coef&, const d2* v, d2* w, double* acc, uint64_t gidx0) const;
template <class Op>
struct is_pairwise {
    template <class T> static constexpr bool test(decltype(T::PAIRWISE)*) { return T::PAIRWISE; }
    template <class T> static constexpr bool test(...) { return false; }
    static constexpr bool value = test<Op>(nullptr);
};

template <class Op>
__device__ __forceinline__ void do_pair(const Op& op, const typename Op::Coef& cf, const d2* v, d2* w, double* acc,
                                        uint64_t gidx0) {
    if constexpr (is_pairwise<Op>::value) {
        op.pair(cf, v, w, acc, gidx0);
    } else {
        double a[Op::NIN ? Op::NIN : 1], b[Op::NIN ? Op::NIN : 1];
        double wa[Op::NOUT ? Op::NOUT : 1], wb[Op::NOUT ? Op::NOUT : 1];
#pragma unroll
        for (int s = 0; s < Op::NIN; ++s) { a[s] = v[s].x; b[s] = v[s].y; }
        op.elem(cf, a, wa, acc, gidx0);
        op.elem(cf, b, wb, acc, gidx0 + 1);
#pragma unroll
        for (int s = 0; s < Op::NOUT; ++s) { w[s].x = wa[s]; w[s].y = wb[s]; }
    }
}

template <class Op>
__global__ __launch_bounds__(BLOCK) void stream_kernel(const Op op, const uint64_t n, const uint64_t gofs,
                                                        const RedCtl red) {
    constexpr int NIN = Op::NIN, NOUT = Op::NOUT, NRED = Op::NRED;
    const typename Op::Coef cf = op.setup();
    double acc[NRED ? NRED : 1];
#pragma unroll
    for (int k = 0; k < (NRED ? NRED : 1); ++k) acc[k] = 0.0;

    const uint64_t n2 = n >> 1;                           // 16-byte pairs
    const uint64_t nch = (n2 + BLOCK - 1) / BLOCK;        // chunks of BLOCK pairs (4 KiB per stream)
    const uint64_t c0 = nch * blockIdx.x / gridDim.x;     // this workgroup's contiguous chunk range
    const uint64_t c1 = nch * (blockIdx.x + 1) / gridDim.x;
    const uint64_t tid = threadIdx.x;

    uint64_t c = c0;
    // fast path: UNROLL full chunks, all loads issued before the first use
    for (; c + UNROLL <= c1 && (c + UNROLL) * BLOCK <= n2; c += UNROLL) {
        d2 v[UNROLL][NIN ? NIN : 1];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t p = (c + u) * BLOCK + tid;
#pragma unroll
            for (int s = 0; s < NIN; ++s) v[u][s] = reinterpret_cast<const d2*>(op.in[s])[p];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t p = (c + u) * BLOCK + tid;
            d2 w[NOUT ? NOUT : 1];
            do_pair<Op>(op, cf, v[u], w, acc, gofs + 2 * p);
#pragma unroll
            for (int s = 0; s < NOUT; ++s) reinterpret_cast<d2*>(op.out[s])[p] = w[s];
        }
    }
    // remainder chunks (at most UNROLL-1 full ones plus the ragged last chunk)
    for (; c < c1; ++c) {
        const uint64_t p = c * BLOCK + tid;
        if (p < n2) {
            d2 v[NIN ? NIN : 1], w[NOUT ? NOUT : 1];
#pragma unroll
            for (int s = 0; s < NIN; ++s) v[s] = reinterpret_cast<const d2*>(op.in[s])[p];
            do_pair<Op>(op, cf, v, w, acc, gofs + 2 * p);
#pragma unroll
            for (int s = 0; s < NOUT; ++s) reinterpret_cast<d2*>(op.out[s])[p] = w[s];
        }
    }
    // odd n: the last element, scalar (never taken by PAIRWISE ops: their n is even)
    if constexpr (!is_pairwise<Op>::value) {
        if ((n & 1) && blockIdx.x == gridDim.x - 1 && tid == 0) {
            const uint64_t e = n - 1;
            double a[NIN ? NIN : 1], wa[NOUT ? NOUT : 1];
#pragma unroll
            for (int s = 0; s < NIN; ++s) a[s] = op.in[s][e];
            op.elem(cf, a, wa, acc, gofs + e);
#pragma unroll
            for (int s = 0; s < NOUT; ++s) op.out[s][e] = wa[s];
        }
    }
    if constexpr (NRED > 0) grid_reduce<NRED>(reinterpret_cast<double(&)[NRED]>(acc), red);
}

}  // namespace lh
