// tools/handoff_bench.hip -- what does a chip-wide hand-off of one f64 sum cost inside a persistent kernel, and how much of
// it can streaming loads that are in flight ACROSS it hide?  (The regime of resident.h's two-loop kernel: 256 workgroups,
// one per CU, each step ends with every workgroup needing the sum of all workgroups' partials.)
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Irust-lbfgs_amd/csrc tools/handoff_bench.hip -o tools/bin/handoff_bench
//   tools/bin/handoff_bench [iters]
//
// Protocols (template V):
//   -1  no hand-off at all: only the streaming loads (the baseline the others are priced against)
//    0  resident.h as of round 2: ds_bpermute wave trees, one publisher per workgroup, EVERY thread polls one workgroup's
//       tagged granules with vector loads, second block reduction, LDS broadcast (five barriers)
//    1  the same with DPP wave sums and two barriers (round 3's first step)
//    2  two levels, polled with SCALAR loads (s_load_dwordx16 glc on uncached memory): 16 leader waves add 16 partials
//       each and publish group totals; every WAVE of every workgroup then reads the 16 group totals itself -- no LDS
//       broadcast, and, the point of it, no vector load in the polling path: a wave's vector loads return in order
//       (vmcnt), so a vector poll cannot be read before every streaming load issued ahead of it has come back, while
//       scalar loads are counted separately (lgkmcnt).
//    3  round 3's form (resident.h today): DPP wave sums, a barrier between the publish and the polls;   4  ... one polling wave
//    5  round 4: two levels with VECTOR polls -- per XCD (workgroup mod 8) by that XCD's leader, then the 8 XCD totals
// K = 16-byte streaming loads per thread that are issued BEFORE the hand-off and consumed after it (resident.h: the window
// of the next step's operands).  Every spin is bounded; a protocol that reads stale data ends with err != 0, not a hang.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "stream.h"
using namespace lh;

typedef unsigned int u16v __attribute__((ext_vector_type(16)));

struct Args {
    unsigned long long* part;  // [2][256][2] tagged granules: a workgroup's partial
    unsigned long long* grp;   // [2][16][2]  tagged granules: a group's total
    const double* stream;
    unsigned long long stream_pairs;
    double* sink;
    unsigned int* err;
    long long* ticks;
    int iters;
};

constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ void publish(unsigned long long* g, const unsigned tag, const double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
    __hip_atomic_store(g, t | (b & 0xffffffffULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, t | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// four s_load_dwordx16 (256 bytes: sixteen 16-byte granule pairs) from a wave-uniform address, not through the scalar cache
__device__ __forceinline__ void sload256(const void* p, u16v& a, u16v& b, u16v& c, u16v& d) {
    asm volatile(
        "s_load_dwordx16 %0, %4, 0x0 glc\n\t"
        "s_load_dwordx16 %1, %4, 0x40 glc\n\t"
        "s_load_dwordx16 %2, %4, 0x80 glc\n\t"
        "s_load_dwordx16 %3, %4, 0xc0 glc\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
        : "s"(p)
        : "memory");
}
// sixteen tagged doubles in 64 dwords [lo, tag, hi, tag] x 16: all tags right?  their sum in index order
__device__ __forceinline__ bool take16(const u16v (&q)[4], const unsigned tag, const int count, double& sum) {
    bool ok = true;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const u16v& v = q[i >> 2];
        const int o = (i & 3) * 4;
        if (i < count) {
            ok = ok && v[o + 1] == tag && v[o + 3] == tag;
            s += __hiloint2double((int)v[o + 2], (int)v[o]);
        }
    }
    sum = s;
    return ok;
}

template <int V>
__device__ __forceinline__ double exchange(double x, const Args& a, const unsigned tag, const int parity, double (*rows)[WAVES],
                                           double* s_tot) {
    const unsigned G = gridDim.x, B = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if constexpr (V == 0) {
        double acc[1] = {x};
        block_sum<1>(acc, rows);
        if (tid == 0) publish(a.part + ((size_t)parity * 256 + B) * 2, tag, acc[0]);
        __syncthreads();
        double tot[1] = {0.0};
        for (unsigned b = tid; b < G; b += BLOCK) {
            const unsigned long long* g = a.part + ((size_t)parity * 256 + b) * 2;
            unsigned long long lo, hi;
            unsigned spins = 0;
            for (;;) {
                lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
                if (++spins > SPIN_MAX) { atomicExch(a.err, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            tot[0] += __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffULL)));
        }
        block_sum<1>(tot, rows);
        if (tid == 0) s_tot[0] = tot[0];
        __syncthreads();
        const double r = s_tot[0];
        __syncthreads();
        return r;
    } else if constexpr (V == 1) {
        double w = wave_sum_dpp(x);
        if (lane == 0) rows[0][wave] = w;
        __syncthreads();
        const double p = ((rows[0][0] + rows[0][1]) + rows[0][2]) + rows[0][3];
        if (tid == 0) publish(a.part + ((size_t)parity * 256 + B) * 2, tag, p);
        double t = 0.0;
        for (unsigned b = tid; b < G; b += BLOCK) {
            const unsigned long long* g = a.part + ((size_t)parity * 256 + b) * 2;
            unsigned long long lo, hi;
            unsigned spins = 0;
            for (;;) {
                lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
                if (++spins > SPIN_MAX) { atomicExch(a.err, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            t += __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffULL)));
        }
        w = wave_sum_dpp(t);
        if (lane == 0) rows[4][wave] = w;
        __syncthreads();
        return ((rows[4][0] + rows[4][1]) + rows[4][2]) + rows[4][3];
    } else if constexpr (V == 3 || V == 4) {
        // 3: V1 with a barrier between the publish and the polls (nobody polls before its own workgroup has published)
        // 4: ... and only wave 0 polls (four partials per lane): a quarter of the requests on the 32 hot cache lines
        double w = wave_sum_dpp(x);
        if (lane == 0) rows[0][wave] = w;
        __syncthreads();
        const double p = ((rows[0][0] + rows[0][1]) + rows[0][2]) + rows[0][3];
        if (tid == 0) publish(a.part + ((size_t)parity * 256 + B) * 2, tag, p);
        __syncthreads();
        if (V == 3 || wave == 0) {
            double t = 0.0;
            const unsigned stride = V == 3 ? BLOCK : 64u;
            for (unsigned b = V == 3 ? tid : (unsigned)lane; b < G; b += stride) {
                const unsigned long long* g = a.part + ((size_t)parity * 256 + b) * 2;
                unsigned long long lo, hi;
                unsigned spins = 0;
                for (;;) {
                    lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
                    if (++spins > SPIN_MAX) { atomicExch(a.err, 1u); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                t += __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffULL)));
            }
            w = wave_sum_dpp(t);
            if (lane == 0) rows[4][wave] = w;
        }
        __syncthreads();
        if constexpr (V == 3) return ((rows[4][0] + rows[4][1]) + rows[4][2]) + rows[4][3];
        return rows[4][0];
    } else if constexpr (V == 5) {
        // round 4: TWO LEVELS with vector polls -- the workgroups of an XCD (B mod 8) first, by wave 0 of that XCD's leader
        // workgroup (32 polls instead of 256), then every workgroup's wave 0 reads the 8 XCD totals (8 polls).  256 x 256 polls
        // become 8 x 32 + 256 x 8 -- but a total now takes two dependent trips through memory.
        double w = wave_sum_dpp(x);
        if (lane == 0) rows[0][wave] = w;
        __syncthreads();
        const double p = ((rows[0][0] + rows[0][1]) + rows[0][2]) + rows[0][3];
        if (tid == 0) publish(a.part + ((size_t)parity * 256 + B) * 2, tag, p);
        __syncthreads();
        const unsigned X = 8u;
        auto poll = [&](const unsigned long long* g, unsigned code) {
            unsigned long long lo, hi;
            unsigned spins = 0;
            for (;;) {
                lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
                if (++spins > SPIN_MAX) { atomicExch(a.err, code); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            return __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffULL)));
        };
        if (B < X && wave == 0) {  // the leader of XCD B: its members are B, B + 8, B + 16, ...
            double t = 0.0;
            for (unsigned i = (unsigned)lane; B + X * i < G; i += 64u) t += poll(a.part + ((size_t)parity * 256 + B + X * i) * 2, 5u);
            const double gs = wave_sum_dpp(t);
            if (lane == 0) publish(a.grp + ((size_t)parity * 16 + B) * 2, tag, gs);
        }
        if (wave == 0) {
            double t = 0.0;
            if ((unsigned)lane < min(X, G)) t = poll(a.grp + ((size_t)parity * 16 + lane) * 2, 6u);
            const double tt = wave_sum_dpp(t);
            if (lane == 0) rows[4][0] = tt;
        }
        __syncthreads();
        return rows[4][0];
    } else if constexpr (V == 2) {
        double w = wave_sum_dpp(x);
        if (lane == 0) rows[0][wave] = w;
        __syncthreads();
        const double p = ((rows[0][0] + rows[0][1]) + rows[0][2]) + rows[0][3];
        if (tid == 0) publish(a.part + ((size_t)parity * 256 + B) * 2, tag, p);
        const unsigned ngroups = (G + 15u) / 16u;
        if ((B & 15u) == 0u && wave == 0) {  // a leader wave: the 16 partials of its group (scalar polls)
            const int count = (int)min(16u, G - B);
            const void* src = a.part + ((size_t)parity * 256 + B) * 2;
            double gs = 0.0;
            unsigned spins = 0;
            for (;;) {
                u16v q[4];
                sload256(src, q[0], q[1], q[2], q[3]);
                if (take16(q, tag, count, gs)) break;
                if (++spins > SPIN_MAX) { atomicExch(a.err, 2u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) publish(a.grp + ((size_t)parity * 16 + (B >> 4)) * 2, tag, gs);
        }
        // every wave for itself: the group totals
        double tot = 0.0;
        unsigned spins = 0;
        const void* src = a.grp + (size_t)parity * 16 * 2;
        for (;;) {
            u16v q[4];
            sload256(src, q[0], q[1], q[2], q[3]);
            if (take16(q, tag, (int)ngroups, tot)) break;
            if (++spins > SPIN_MAX) { atomicExch(a.err, 3u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        return tot;
    } else {
        return x;
    }
}

template <int V, int K>
__global__ __launch_bounds__(BLOCK) void hb_kernel(const Args a) {
    extern __shared__ char pad[];  // (dynamic LDS only to keep one workgroup per CU)
    __shared__ double rows[8][WAVES];
    __shared__ double s_tot[1];
    const unsigned G = gridDim.x, B = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) pad[0] = 0;
    double sink = 0.0;
    unsigned tag = 1;
    int parity = 0;
    const long long t0 = wall_clock64();
    const double want_unit = 0.5 * (double)G * (double)(G + 1);
    for (int it = 0; it < a.iters; ++it) {
        d2 w[K > 0 ? K : 1];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned int p = (((unsigned int)it * K + k) * G + B) * BLOCK + tid;   // (wraps: any address will do)
            w[k] = ld16<true>(a.stream, p & (unsigned int)(a.stream_pairs - 1));
        }
        asm volatile("" ::: "memory");
        const double x = tid == 0 ? (double)(B + 1) * (double)(it + 1) : 0.0;
        double total = x;
        if constexpr (V >= 0) {
            total = exchange<V>(x, a, tag, parity, rows, s_tot);
            if (total != want_unit * (double)(it + 1)) atomicExch(a.err, 9u);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            asm volatile("" : "+v"(w[k]));  // (the loaded values may only be USED from here on: they fly across the hand-off)
            sink += w[k].x + w[k].y;
        }
        sink += total;
        tag += 1;
        parity ^= 1;
    }
    const long long t1 = wall_clock64();
    if (tid == 0) {
        a.ticks[B] = t1 - t0;
        a.sink[B] = sink;
    }
}

template <int V, int K>
double run(Args a, int grid, const char* what) {
    hipMemset(a.part, 0, 2 * 256 * 2 * 8);
    hipMemset(a.grp, 0, 2 * 16 * 2 * 8);
    hipMemset(a.err, 0, 4);
    auto kern = hb_kernel<V, K>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), 100 * 1024, 0, a);
    hipDeviceSynchronize();
    std::vector<long long> t(grid);
    unsigned err = 0;
    hipMemcpy(t.data(), a.ticks, grid * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (long long v : t) mx = v > mx ? v : mx;
    const double us = mx * 0.01 / a.iters;
    printf("%-34s V=%2d K=%2d grid=%3d: %7.3f us per iteration%s\n", what, V, K, grid, us, err ? "   *** ERROR (stale data or timeout) ***" : "");
    fflush(stdout);
    return us;
}

template <int K>
void sweep(Args a, int grid, const char* mem, bool uc) {
    char buf[64];
    snprintf(buf, sizeof(buf), "streaming only");
    const double base = run<-1, K>(a, grid, buf);
    snprintf(buf, sizeof(buf), "round-2 hand-off (%s)", mem);
    const double v0 = run<0, K>(a, grid, buf);
    snprintf(buf, sizeof(buf), "DPP + 2 barriers (%s)", mem);
    const double v1 = run<1, K>(a, grid, buf);
    snprintf(buf, sizeof(buf), "two-level scalar polls (%s)", mem);
    const double v2 = uc ? run<2, K>(a, grid, buf) : 0.0;   // (scalar loads see stale lines of cached memory: not run there)
    snprintf(buf, sizeof(buf), "DPP, barrier before polls (%s)", mem);
    const double v3 = run<3, K>(a, grid, buf);
    snprintf(buf, sizeof(buf), "... only wave 0 polls (%s)", mem);
    const double v4 = run<4, K>(a, grid, buf);
    snprintf(buf, sizeof(buf), "two levels, vector polls (%s)", mem);
    const double v5 = run<5, K>(a, grid, buf);
    printf("   => visible cost of a hand-off with %2d loads in flight: round-2 %.2f, DPP/no barrier %.2f, scalar %.2f, DPP+barrier %.2f, "
           "one polling wave %.2f, two levels (per XCD, then 8 leaders) %.2f us\n", K, v0 - base, v1 - base, v2 - base, v3 - base, v4 - base,
           v5 - base);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    Args a{};
    a.iters = iters;
    a.stream_pairs = (1ull << 30) / 16;  // 1 GiB
    hipMalloc((void**)&a.stream, a.stream_pairs * 16);
    hipMemset((void*)a.stream, 0, a.stream_pairs * 16);
    hipMalloc((void**)&a.sink, 4096 * 8);
    hipMalloc((void**)&a.ticks, 4096 * 8);
    hipMalloc((void**)&a.err, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int uc = 1; uc >= 0; --uc) {
        void *p = nullptr, *g = nullptr;
        if (uc) {
            hipExtMallocWithFlags(&p, 2 * 256 * 2 * 8, hipDeviceMallocUncached);
            hipExtMallocWithFlags(&g, 4096, hipDeviceMallocUncached);
        } else {
            hipMalloc(&p, 2 * 256 * 2 * 8);
            hipMalloc(&g, 4096);
        }
        a.part = (unsigned long long*)p;
        a.grp = (unsigned long long*)g;
        const char* mem = uc ? "uncached memory" : "plain hipMalloc";
        printf("---- granules in %s, %d CUs\n", mem, cus);
        for (int grid : {cus}) {
            sweep<0>(a, grid, mem, uc);
            sweep<8>(a, grid, mem, uc);
            sweep<16>(a, grid, mem, uc);
            sweep<24>(a, grid, mem, uc);
        }
        hipFree(p);
        hipFree(g);
    }
    return 0;
}
