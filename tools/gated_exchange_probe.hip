// tools/gated_exchange_probe.hip -- EXPERIMENT: can a persistent kernel that fills the chip hand a few scalars to kernels that the
// host enqueued ahead on ANOTHER stream (a stand-in for ncclAllReduce, which only exists as a host-enqueued kernel), and get the
// result back, without leaving the chip?  And what does one such round trip cost?
//
//   stream A: ONE persistent kernel of (CUs - reserve) workgroups, one per CU (it asks for all of a CU's LDS).  Workgroup 0 makes
//             N exchanges: stores its scalars into an uncached buffer, raises flag A = seq, spins until flag B == seq, reads the
//             buffer back; the other workgroups wait for the kernel's end the way the resident two-loop's workgroups wait in a
//             hand-off.
//   stream B: gate(seq 1) -> reduce -> [post(seq k) + gate(seq k+1)] -> reduce -> ... -> post(seq N): all enqueued before anything
//             runs.  gate = one wave spinning on flag A; reduce = a small kernel that changes the buffer (the all-reduce's
//             stand-in: buf[i] = 2 * buf[i] + 1); post = flag B = seq.
// Prints the round trip per exchange as workgroup 0 sees it.   hipcc --offload-arch=gfx950 -O2 tools/gated_exchange_probe.hip -o probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                  \
    do {                                                                          \
        hipError_t e_ = (e);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr unsigned long long TIMEOUT = 200000000ull;  // 2 s of the 100 MHz wall clock: every spin is bounded

struct Flags {
    unsigned long long a, b, done, err;
};

__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void persistent(Flags* f, double* buf, int n_exchanges, int nvals, unsigned long long* ticks,
                                                  double* check) {
    extern __shared__ char lds[];  // (all of the CU's LDS: one workgroup per CU, like the resident two-loop)
    (void)lds;
    if (blockIdx.x != 0) {  // the other workgroups: wait for the end, bounded
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (ld(&f->done) == 0 && (unsigned long long)(wall_clock64() - t0) < TIMEOUT * 4) __builtin_amdgcn_s_sleep(8);
        }
        return;
    }
    double acc = 0.0;
    for (int x = 1; x <= n_exchanges; ++x) {
        const long long t0 = wall_clock64();
        if ((int)threadIdx.x < nvals) {
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(buf + threadIdx.x), __double_as_longlong((double)(x + threadIdx.x)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the values have left this CU before the flag does
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            st(&f->a, (unsigned long long)x);
            while (ld(&f->b) != (unsigned long long)x) {
                if ((unsigned long long)(wall_clock64() - t0) > TIMEOUT) {
                    st(&f->err, 1ull);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nvals) {
            const double v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(buf + threadIdx.x),
                                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            if (v != 2.0 * (double)(x + threadIdx.x) + 1.0) st(&f->err, 2ull);
            acc += v;
        }
        if (threadIdx.x == 0) ticks[x - 1] = (unsigned long long)(wall_clock64() - t0);
        if (ld(&f->err) != 0) break;
    }
    if (threadIdx.x == 0) {
        *check = acc;
        st(&f->done, 1ull);
    }
}

// post (flag B = post_seq, 0 = none) then gate (spin until flag A == gate_seq, 0 = none)
__global__ __launch_bounds__(64) void post_gate(Flags* f, unsigned long long post_seq, unsigned long long gate_seq) {
    if (threadIdx.x != 0) return;
    if (post_seq) st(&f->b, post_seq);
    if (gate_seq) {
        const long long t0 = wall_clock64();
        while (ld(&f->a) != gate_seq) {
            if ((unsigned long long)(wall_clock64() - t0) > TIMEOUT || ld(&f->err) != 0) {
                st(&f->err, 3ull);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
}
__global__ __launch_bounds__(256) void reduce_standin(double* buf, int nvals) {
    if ((int)threadIdx.x < nvals) buf[threadIdx.x] = 2.0 * buf[threadIdx.x] + 1.0;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200, reserve = argc > 2 ? atoi(argv[2]) : 8, nvals = 4;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, grid = cus - reserve;
    Flags* f;
    double *buf, *check;
    unsigned long long* ticks;
    CHECK(hipExtMallocWithFlags((void**)&f, sizeof(Flags), hipDeviceMallocUncached));
    CHECK(hipExtMallocWithFlags((void**)&buf, 64 * sizeof(double), hipDeviceMallocUncached));
    CHECK(hipMalloc(&check, sizeof(double)));
    CHECK(hipMalloc(&ticks, N * sizeof(unsigned long long)));
    CHECK(hipMemset(f, 0, sizeof(Flags)));
    CHECK(hipMemset(buf, 0, 64 * sizeof(double)));
    hipStream_t a, b;
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const size_t lds = 150 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int order = 0; order < 2; ++order) {  // chain first / persistent kernel first
        CHECK(hipMemset(f, 0, sizeof(Flags)));
        CHECK(hipDeviceSynchronize());
        auto chain = [&] {
            post_gate<<<1, 64, 0, b>>>(f, 0ull, 1ull);
            for (int x = 1; x <= N; ++x) {
                reduce_standin<<<1, 256, 0, b>>>(buf, nvals);
                post_gate<<<1, 64, 0, b>>>(f, (unsigned long long)x, x < N ? (unsigned long long)(x + 1) : 0ull);
            }
        };
        if (order == 0) chain();
        persistent<<<grid, 256, lds, a>>>(f, buf, N, nvals, ticks, check);
        if (order == 1) chain();
        CHECK(hipStreamSynchronize(a));
        CHECK(hipStreamSynchronize(b));
        Flags hf;
        double hc;
        std::vector<unsigned long long> ht(N);
        CHECK(hipMemcpy(&hf, f, sizeof(hf), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&hc, check, sizeof(hc), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(ht.data(), ticks, N * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum = 0.0, mx = 0.0, mn = 1e30;
        for (int i = 10; i < N; ++i) {
            const double us = ht[i] * 0.01;
            sum += us;
            mx = us > mx ? us : mx;
            mn = us < mn ? us : mn;
        }
        printf("%s: %d CUs, persistent grid %d (reserve %d), %d exchanges of %d doubles: err %llu, round trip per exchange mean %.2f us "
               "(min %.2f, max %.2f; first %.1f us)\n",
               order == 0 ? "chain enqueued first" : "persistent kernel first", cus, grid, reserve, N, nvals, hf.err, sum / (N - 10), mn, mx,
               ht[0] * 0.01);
        if (hf.err) return 1;
    }
    return 0;
}
