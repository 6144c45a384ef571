// tools/tune_stream.hip -- sweep launch geometry / unroll / cache hints of the streaming skeleton
// on the dominant kernel (two-loop step, 3 reads + 1 write) and a few others.  Development tool:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 tools/tune_stream.hip -o gpurun_out/tune && gpurun_out/tune [n]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../rust-lbfgs_amd/csrc/ops.h"
#include "../rust-lbfgs_amd/csrc/gram.h"

using namespace lh;

// EXPERIMENT (not adopted, see DESIGN.md section 7 and profiles/r01_tune_peek_pair.log): operators for issuing the
// recursion two steps at a time.  They live here, not in ops.h, because the product does not use them.
namespace lh {
// ---- the recursion two steps at a time: "peek, then apply both" -------------------------------------------
// A write pass costs 1.67x a read pass on MI355X (0.183 ms vs 0.110 ms per 0.8 GB, fitted on every kernel shape of
// profiles/r01_bench_n1e8_m10.md).  Consecutive steps a, b of the recursion
//     q_a = q + c_a*u_a ; dot_a = w_a.q_a        q_b = q_a + c_b*u_b ; dot_b = w_b.q_b
// where c_b needs the GLOBAL sum dot_a are therefore issued as
//     peek  : dot_a over q + c_a*u_a, q_a NOT written                                    (3r 0w)
//     pair  : q_b = (q + c_a*u_a) + c_b*u_b written once, dot_b                          (4r 1w)
// = 7r 1w per two steps instead of 6r 2w: the same element-wise arithmetic in the same order (q_a is recomputed, to
// the same bits, never re-associated), one write pass fewer.
//
// One step of the recursion as the kernels see it (lbfgs.rs:582-601):
struct TwoLoopStepRef {
    const double* dot_in;  // numerator produced by the previous reduction (global sum)
    const double* ys_j;    // ys of the slot whose coefficient this step applies
    double* alpha_j;       // alpha of that slot (stored in mode A, read in mode B)
    int mode_b;            // 0: first loop  c = -(dot/ys), alpha := dot/ys      1: second loop  c = alpha - dot/ys
    double sign;           // -1.0 when the source is g and the recursion starts from d = -g (core.rs:95-101), else 1.0
    int scale;             // 1: this step is followed by q *= gamma (lbfgs.rs:591), gamma = board ratio ys/yy
};
__device__ __forceinline__ double two_loop_coef(const TwoLoopStepRef& r, bool store_alpha) {
    const double q = *r.dot_in / *r.ys_j;
    if (r.mode_b) return *r.alpha_j - q;
    if (store_alpha && blockIdx.x == 0 && threadIdx.x == 0) *r.alpha_j = q;
    return -q;
}
struct TwoLoopPairCoef {
    double sign, c1, g1, c2, g2;
};

// peek: the sum a step would produce, without writing its vector.  WMODE 0: w is a third stream; 1: w = u (the
// gamma transition).  x*1.0 and x*(+-1.0) are exact, so the optional scale and sign are plain multiplications.
template <int WMODE>
struct OpTwoLoopPeek {
    static constexpr int NIN = (WMODE == 1) ? 2 : 3, NOUT = 0, NRED = 1;
    static constexpr int TUNE_MAP = 1, TUNE_UNROLL = 2;
    const double* in[3];  // src, u, w
    double* out[1];
    TwoLoopStepRef a;
    const double* gamma_num;
    const double* gamma_den;
    typedef TwoLoopPairCoef Coef;
    __device__ Coef setup() const {
        Coef cf;
        cf.sign = a.sign;
        cf.c1 = two_loop_coef(a, true);
        cf.g1 = a.scale ? (*gamma_num / *gamma_den) : 1.0;
        cf.c2 = 0.0;
        cf.g2 = 1.0;
        return cf;
    }
    __device__ void elem(const Coef& cf, const double* v, double*, double* acc, uint64_t) const {
        double q = cf.sign * v[0] + cf.c1 * v[1];  // math.rs:35
        q = q * cf.g1;                             // math.rs:47 (exact no-op when g1 == 1)
        acc[0] += ((WMODE == 1) ? v[1] : v[2]) * q;
    }
};

// pair: apply step a (whose sum the peek produced) and step b, write once, produce b's sum(s).
//   VMODE 0: w_b is a fourth stream (4r 1w); 1: w_b = u_b (b is the gamma transition, 3r 1w);
//   VMODE 2: b is the last step: ||d||^2 and g.d with g as the fourth stream; 3: the same under OWL-QN (see OpTwoLoopStep)
template <int VMODE>
struct OpTwoLoopPair {
    static constexpr int NIN = (VMODE == 1) ? 3 : 4, NOUT = 1, NRED = (VMODE == 2) ? 2 : (VMODE == 3 ? 4 : 1);
    static constexpr int TUNE_MAP = 1, TUNE_UNROLL = 2;
    const double* in[4];  // src, u_a, u_b, w_b (VMODE 2/3: g / pg)
    double* out[1];       // dst (= d)
    TwoLoopStepRef a, b;
    const double* gamma_num;
    const double* gamma_den;
    uint64_t owl_start, owl_end;  // VMODE 3 only
    typedef TwoLoopPairCoef Coef;
    __device__ Coef setup() const {
        Coef cf;
        const double gamma = (a.scale || b.scale) ? (*gamma_num / *gamma_den) : 1.0;  // lbfgs.rs:691 ys/yy
        cf.sign = a.sign;
        cf.c1 = two_loop_coef(a, false);  // alpha_a was stored by the peek
        cf.g1 = a.scale ? gamma : 1.0;
        cf.c2 = two_loop_coef(b, true);
        cf.g2 = b.scale ? gamma : 1.0;
        return cf;
    }
    __device__ void elem(const Coef& cf, const double* v, double* w, double* acc, uint64_t gi) const {
        double q = cf.sign * v[0] + cf.c1 * v[1];
        q = q * cf.g1;
        q = q + cf.c2 * v[2];
        q = q * cf.g2;
        const double wv = (VMODE == 0) ? v[3] : (VMODE == 1 ? v[2] : q);
        acc[0] += wv * q;
        if constexpr (VMODE == 2) acc[1] += v[3] * q;  // g.d
        if constexpr (VMODE == 3) {
            if (gi >= owl_start && gi < owl_end && signum0(q) != signum0(-v[3])) q = 0.0;
            acc[2] += q * q;
            acc[3] += v[3] * q;                        // pg.d (core.rs:90)
        }
        w[0] = q;
    }
};

}  // namespace lh

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bufs {
    double *q, *u, *v, *board, *partials;
    unsigned* ticket;
    unsigned long long* gran;
    unsigned* err;
    uint64_t n;
};
static DevCounters* g_ctr = nullptr;
// the hand-off part of a RedCtl for the next launch (the tag comes from device counters, as in the library)
static void handoff(RedCtl& red, const Bufs& b) {
    if (!g_ctr) {
        DevCounters init{1u, 1u, 0ull};
        CK(hipMalloc(&g_ctr, sizeof(DevCounters)));
        CK(hipMemcpy(g_ctr, &init, sizeof(init), hipMemcpyHostToDevice));
    }
    red.partials = b.partials; red.ticket = b.ticket; red.gran = b.gran; red.err = b.err;
    red.timeout_ticks = 200000000ULL;
    red.ctr = g_ctr;
    red.tagged = 1u;
}

template <class Op, int UNR, unsigned NTL, unsigned NTS, int MAP, int SPAN = 1>
float run(const Op& op, const Bufs& b, int grid, int reps, int* occ) {
    RedCtl red{};
    for (int k = 0; k < RED_PTRS; ++k) red.out[k] = b.board + 2 + k;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, stream_kernel<Op, UNR, NTL, NTS, MAP, SPAN>, BLOCK, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) { handoff(red, b); hipLaunchKernelGGL((stream_kernel<Op, UNR, NTL, NTS, MAP, SPAN>), dim3(grid), dim3(BLOCK), 0, 0, op, b.n, 0, red); }
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        handoff(red, b);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stream_kernel<Op, UNR, NTL, NTS, MAP, SPAN>), dim3(grid), dim3(BLOCK), 0, 0, op, b.n, 0, red);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ts[ts.size() / 2];
}

template <int UNR, unsigned NTL, unsigned NTS, int MAP, int SPAN = 1>
void sweep_step(const Bufs& b, const std::vector<int>& grids) {
    OpTwoLoopStep<false, false, 0> op{};
    op.in[0] = b.q; op.in[1] = b.u; op.in[2] = b.v; op.out[0] = b.q;
    op.dot_in = b.board; op.ys_j = b.board + 1; op.alpha_j = b.board + 10; op.gamma_num = b.board; op.gamma_den = b.board + 1;
    op.mode_b = 1;
    for (int g : grids) {
        int occ = 0;
        float ms = run<decltype(op), UNR, NTL, NTS, MAP, SPAN>(op, b, g, 15, &occ);
        printf("step3r1w_span%d map=%d unroll=%d ntl=%d nts=%d grid=%5d (%.2f/CU, occ=%d) : %8.3f ms  %7.1f GB/s\n", SPAN, MAP, UNR, (int)(NTL & 15u), (int)(NTS & 15u), g,
               g / 256.0, occ, ms, 32.0 * b.n / ms / 1e6);
    }
}

template <int UNR, unsigned NTL, unsigned NTS, int MAP>
void sweep_copy(const Bufs& b, const std::vector<int>& grids) {
    OpCopy<false> op{};
    op.in[0] = b.u; op.out[0] = b.v;
    for (int g : grids) {
        int occ = 0;
        float ms = run<decltype(op), UNR, NTL, NTS, MAP>(op, b, g, 15, &occ);
        printf("copy1r1w map=%d unroll=%d ntl=%d nts=%d grid=%5d (%.2f/CU, occ=%d) : %8.3f ms  %7.1f GB/s\n", MAP, UNR, (int)(NTL & 15u), (int)(NTS & 15u), g,
               g / 256.0, occ, ms, 16.0 * b.n / ms / 1e6);
    }
}

template <int UNR, unsigned NTL, int MAP>
void sweep_dot(const Bufs& b, const std::vector<int>& grids) {
    OpDot op{};
    op.in[0] = b.u; op.in[1] = b.v;
    for (int g : grids) {
        int occ = 0;
        float ms = run<decltype(op), UNR, NTL, 0u, MAP>(op, b, g, 15, &occ);
        printf("dot2r    map=%d unroll=%d ntl=%d nts=0 grid=%5d (%.2f/CU, occ=%d) : %8.3f ms  %7.1f GB/s\n", MAP, UNR, (int)(NTL & 15u), g, g / 256.0, occ,
               ms, 16.0 * b.n / ms / 1e6);
    }
}

template <int UNR, unsigned NTL, unsigned NTS, int MAP, int SPAN = 1>
void sweep_hist(const Bufs& b, const std::vector<int>& grids, double* extra[4]) {
    OpHistUpdate<false> op{};
    op.in[0] = b.q; op.in[1] = b.u; op.in[2] = b.v; op.in[3] = extra[0]; op.out[0] = extra[1]; op.out[1] = extra[2];
    for (int g : grids) {
        int occ = 0;
        float ms = run<decltype(op), UNR, NTL, NTS, MAP, SPAN>(op, b, g, 15, &occ);
        printf("hist4r2w_s%d map=%d unroll=%d ntl=%d nts=%d grid=%5d (%.2f/CU, occ=%d) : %8.3f ms  %7.1f GB/s\n", SPAN, MAP, UNR,
               (int)(NTL & 15u), (int)(NTS & 15u), g, g / 256.0, occ, ms, 48.0 * b.n / ms / 1e6);
    }
}
template <int UNR, unsigned NTL, unsigned NTS, int MAP, int SPAN = 1>
void sweep_lineeval(const Bufs& b, const std::vector<int>& grids, double* extra[4]) {
    OpObjLineEval<ObjQuadratic> op{};
    op.in[0] = b.u; op.in[1] = b.v; op.out[0] = extra[1]; op.out[1] = extra[2]; op.step = 1e-3; op.obj = {1, 2};
    for (int g : grids) {
        int occ = 0;
        float ms = run<decltype(op), UNR, NTL, NTS, MAP, SPAN>(op, b, g, 15, &occ);
        printf("line2r2w_s%d map=%d unroll=%d ntl=%d nts=%d grid=%5d (%.2f/CU, occ=%d) : %8.3f ms  %7.1f GB/s\n", SPAN, MAP, UNR,
               (int)(NTL & 15u), (int)(NTS & 15u), g, g / 256.0, occ, ms, 32.0 * b.n / ms / 1e6);
    }
}

template <int UNR, unsigned NTL, unsigned NTS, int MAP, int SPAN = 1>
void sweep_update_from_step(const Bufs& b, const std::vector<int>& grids, double* extra[4]) {
    OpHistUpdateFromStep<ObjQuadratic, false> op{};
    op.in[0] = b.q; op.in[1] = b.u; op.in[2] = b.v;
    op.out[0] = extra[0]; op.out[1] = extra[1]; op.out[2] = extra[2]; op.out[3] = extra[3];
    op.t = 1e-3; op.neg_step = 0.0; op.obj = {0x5EED0001ULL, 0x5EED0002ULL};
    OpObjLineProbe<ObjQuadratic> pr{};
    pr.in[0] = b.q; pr.in[1] = b.u; pr.step = 1e-3; pr.obj = {0x5EED0001ULL, 0x5EED0002ULL};
    for (int g : grids) {
        int occ = 0;
        float ms = run<decltype(op), UNR, NTL, NTS, MAP, SPAN>(op, b, g, 9, &occ);
        float ms2 = run<decltype(pr), UNR, NTL, NTS, MAP, SPAN>(pr, b, g, 9, &occ);
        printf("upd3r4w_s%d map=%d unroll=%d grid=%5d (%.2f/CU) : %8.3f ms %7.1f GB/s | probe2r : %8.3f ms %7.1f GB/s\n", SPAN, MAP, UNR, g,
               g / 256.0, ms, 56.0 * b.n / ms / 1e6, ms2, 16.0 * b.n / ms2 / 1e6);
    }
}

template <int UNR, unsigned NTL, unsigned NTS, int MAP, int SPAN = 1>
void sweep_combine(const Bufs& b, const std::vector<int>& grids, double** vecs) {
    OpGramCombine<10> op{};
    for (int j = 0; j < 21; ++j) op.in[j] = vecs[j];
    op.out[0] = vecs[21];
    op.delta = b.board + 20;
    for (int g : grids) {
        int occ = 0;
        float ms = run<decltype(op), UNR, NTL, NTS, MAP, SPAN>(op, b, g, 7, &occ);
        printf("comb21r1w_s%d map=%d unroll=%d ntl=%d nts=%d grid=%5d (%.2f/CU, occ=%d) : %8.3f ms  %7.1f GB/s\n", SPAN, MAP, UNR,
               (int)(NTL & 15u), (int)(NTS & 15u), g, g / 256.0, occ, ms, 22.0 * 8 * b.n / ms / 1e6);
    }
}
template <int UNR>
void sweep_rows(const Bufs& b, const std::vector<int>& grids, double** vecs) {
    GramRowsArgs<10> a{};
    for (int j = 0; j < 21; ++j) a.in[j] = vecs[j];
    RedCtl red{};
    red.partials = b.partials; red.ticket = b.ticket; red.out_contig = b.board + 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int g : grids) {
        std::vector<float> ts;
        for (int r = 0; r < 8; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((gram_rows_kernel<10, true, UNR>), dim3(g), dim3(BLOCK), 0, 0, a, b.n, red);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        float ms = ts[ts.size() / 2];
        printf("rows21r_s1 map=1 unroll=%d ntl=15 nts=0 grid=%5d (%.2f/CU, occ=0) : %8.3f ms  %7.1f GB/s\n", UNR, g, g / 256.0, ms,
               21.0 * 8 * b.n / ms / 1e6);
    }
}

template <int UNR, unsigned NTL, unsigned NTS, int MAP, int SPAN = 1>
void sweep_peek_pair(const Bufs& b, const std::vector<int>& grids, double* extra[4]) {
    TwoLoopStepRef ra{b.board, b.board + 1, b.board + 10, 1, 1.0, 0}, rb{b.board, b.board + 1, b.board + 11, 1, 1.0, 0};
    OpTwoLoopPeek<0> pk{};
    pk.in[0] = b.q; pk.in[1] = b.u; pk.in[2] = b.v; pk.a = ra; pk.gamma_num = b.board; pk.gamma_den = b.board + 1;
    OpTwoLoopPair<0> pr{};
    pr.in[0] = b.q; pr.in[1] = b.u; pr.in[2] = b.v; pr.in[3] = extra[0]; pr.out[0] = b.q; pr.a = ra; pr.b = rb;
    pr.gamma_num = b.board; pr.gamma_den = b.board + 1;
    for (int g : grids) {
        int occ = 0;
        float ms1 = run<decltype(pk), UNR, NTL, NTS, MAP, SPAN>(pk, b, g, 15, &occ);
        float ms2 = run<decltype(pr), UNR, NTL, NTS, MAP, SPAN>(pr, b, g, 15, &occ);
        printf("peek3r+pair4r1w_s%d map=%d unroll=%d ntl=%d nts=%d grid=%5d (%.2f/CU, occ=%d) : peek %8.3f ms %7.1f GB/s | pair %8.3f ms %7.1f GB/s | both %8.3f ms\n",
               SPAN, MAP, UNR, (int)(NTL & 15u), (int)(NTS & 15u), g, g / 256.0, occ, ms1, 24.0 * b.n / ms1 / 1e6, ms2,
               40.0 * b.n / ms2 / 1e6, ms1 + ms2);
    }
}

// EXPERIMENT (not adopted, profiles/r01_tune_persistent_two_loop.log): grid-wide all-reduce for a persistent kernel.
namespace lh {
// ---- persistent kernels: a grid-wide sum that EVERY workgroup receives ---------------------------------------
// For kernels that stay resident across dependent phases (the persistent two-loop): all workgroups must be
// co-resident (cooperative launch).  Phase `step` (0, 1, 2, ... within one launch): each workgroup publishes its
// partials, signals a monotonic arrival counter, waits until all G have signalled, then sums all partials itself in
// the same fixed order as grid_reduce's last workgroup -- so the totals are bitwise those of the one-launch-per-step
// path.  The partial buffers alternate with the parity of `step`: a workgroup can be at most one phase ahead of the
// slowest one, which may still be reading the previous phase's buffer.  The release/acquire fences at agent scope
// also order the streamed vector itself (a phase may read elements another workgroup wrote in the previous phase,
// and L1 lines of this CU may be stale).  The spin is bounded by a wall-clock timeout that raises *err.
constexpr int PERSIST_MAX_RED = 4;
struct PersistCtl {
    double* partials;              // [2][PERSIST_MAX_RED][MAX_GRID]
    unsigned int* arrive;          // zero at launch
    unsigned int* err;
    unsigned long long timeout_ticks;
};

template <int NRED, bool FENCES = true>
__device__ __forceinline__ void grid_allreduce(double (&acc)[NRED], const PersistCtl& pc, unsigned int step) {
    static_assert(NRED <= PERSIST_MAX_RED, "persistent partial buffer");
    __shared__ double lds[NRED][WAVES];
    __shared__ double s_tot[NRED];
    block_sum<NRED>(acc, lds);
    const unsigned int G = gridDim.x;
    double* buf = pc.partials + (size_t)(step & 1u) * PERSIST_MAX_RED * MAX_GRID;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) store_agent(buf + (size_t)k * MAX_GRID + blockIdx.x, acc[k]);
    }
    if constexpr (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this thread's stores (vector + partials) have left the CU
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(pc.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = G * (step + 1u);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(pc.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if ((unsigned long long)(wall_clock64() - t0) > pc.timeout_ticks) {
                atomicExch(pc.err, 2u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if constexpr (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // nothing cached in this CU predates the barrier
    double tot[NRED];
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
        double t = 0.0;
        for (unsigned int b = threadIdx.x; b < G; b += BLOCK) t += load_agent(buf + (size_t)k * MAX_GRID + b);
        tot[k] = t;
    }
    block_sum<NRED>(tot, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) s_tot[k] = tot[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NRED; ++k) acc[k] = s_tot[k];
    __syncthreads();  // lds / s_tot are reused by the next phase
}

}  // namespace lh

// EXPERIMENT: STEPS dependent two-loop steps inside ONE persistent launch (grid barrier + consumer-side sum per step)
template <int UNR, unsigned NT, int MAP, bool FENCES>
__global__ __launch_bounds__(BLOCK) void persist_probe(OpTwoLoopStep<false, false, 0> op, uint64_t n, PersistCtl pc, int steps,
                                                        double* out) {
    double tot[1] = {1e-12};
    for (int s = 0; s < steps; ++s) {
        TwoLoopCoef cf{tot[0] * 1e-30, 1.0};
        double acc[1] = {0.0};
        stream_body<decltype(op), UNR, NT, NT, MAP, 1>(op, cf, n, 0, acc);
        grid_allreduce<1, FENCES>(acc, pc, (unsigned)s);
        tot[0] = acc[0];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = tot[0];
}

int main(int argc, char** argv) {
    Bufs b{};
    b.n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ULL;
    size_t bytes = b.n * sizeof(double);
    CK(hipMalloc(&b.q, bytes));
    CK(hipMalloc(&b.u, bytes));
    CK(hipMalloc(&b.v, bytes));
    CK(hipMalloc(&b.board, 64 * sizeof(double)));
    CK(hipMalloc(&b.partials, (size_t)MAX_RED * MAX_GRID * sizeof(double)));
    CK(hipMalloc(&b.ticket, 64));
    CK(hipMemset(b.ticket, 0, 64));
    CK(hipMalloc(&b.gran, (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long)));
    CK(hipMemset(b.gran, 0, (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long)));
    CK(hipMalloc(&b.err, 64));
    CK(hipMemset(b.err, 0, 64));
    std::vector<double> h(b.n);
    for (size_t i = 0; i < b.n; ++i) h[i] = 1e-3 * (double)((i * 2654435761ULL) % 1000) - 0.5;
    CK(hipMemcpy(b.q, h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.u, h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.v, h.data(), bytes, hipMemcpyHostToDevice));
    double bd[64];
    for (int i = 0; i < 64; ++i) bd[i] = 1.0;
    bd[0] = 1e-12;  // tiny coefficient so q stays bounded across repetitions
    CK(hipMemcpy(b.board, bd, sizeof(bd), hipMemcpyHostToDevice));

    if (argc > 2 && atoi(argv[2]) == -2) {  // store-policy experiment: build with -DLH_STORE_POLICY=k
        std::vector<int> g1 = {200, 216, 232};
        printf("LH_STORE_POLICY=%d\n", LH_STORE_POLICY);
        sweep_step<2, ~0u, ~0u, 1, 1>(b, g1);
        sweep_copy<4, ~0u, ~0u, 2>(b, g1);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == -7) {  // the two kernels that carry the hashed objective: map / unroll / grid
        double* extra[4];
        for (auto& e : extra) CK(hipMalloc(&e, bytes));
        std::vector<int> g1 = {216, 256, 384, 512, 768, 1024};
        constexpr unsigned ALL = ~0u;
        sweep_update_from_step<4, ALL, ALL, 2, 1>(b, g1, extra);
        sweep_update_from_step<2, ALL, ALL, 2, 1>(b, g1, extra);
        sweep_update_from_step<2, ALL, ALL, 1, 1>(b, g1, extra);
        sweep_update_from_step<1, ALL, ALL, 1, 1>(b, g1, extra);
        sweep_update_from_step<4, ALL, ALL, 1, 1>(b, g1, extra);
        sweep_update_from_step<2, ALL, ALL, 2, 2>(b, g1, extra);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == -5) {  // persistent two-loop probe vs one launch per step
        const int STEPS = 20;
        OpTwoLoopStep<false, false, 0> op{};
        op.in[0] = b.q; op.in[1] = b.u; op.in[2] = b.v; op.out[0] = b.q;
        op.dot_in = b.board; op.ys_j = b.board + 1; op.alpha_j = b.board + 10; op.gamma_num = b.board; op.gamma_den = b.board + 1;
        op.mode_b = 1;
        PersistCtl pc{};
        CK(hipMalloc(&pc.partials, 2 * PERSIST_MAX_RED * MAX_GRID * sizeof(double)));
        CK(hipMalloc(&pc.arrive, 64));
        CK(hipMalloc(&pc.err, 64));
        CK(hipMemset(pc.err, 0, 64));
        pc.timeout_ticks = 200000000ULL;  // 2 s
        RedCtl red{};
        red.partials = b.partials; red.ticket = b.ticket;
        for (int k = 0; k < RED_PTRS; ++k) red.out[k] = b.board + 2 + k;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int grid : {216, 256}) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < STEPS; ++i) {
                    handoff(red, b);
                    hipLaunchKernelGGL((stream_kernel<decltype(op), 2, 0u, 0u, 1, 1>), dim3(grid), dim3(BLOCK), 0, 0, op, b.n, 0, red);
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 2) printf("n=%llu grid=%d: %d launches           : %8.1f us per step\n", (unsigned long long)b.n, grid, STEPS, ms * 1000 / STEPS);
            }
            for (int fences = 0; fences < 2; ++fences)
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(pc.arrive, 0, 4, 0));
                int steps = STEPS;
                uint64_t n = b.n;
                double* out = b.board + 40;
                void* args[] = {&op, &n, &pc, &steps, &out};
                CK(hipEventRecord(e0, 0));
                if (fences) CK(hipLaunchCooperativeKernel((const void*)persist_probe<2, 0u, 1, true>, dim3(grid), dim3(BLOCK), args, 0, 0));
                else CK(hipLaunchCooperativeKernel((const void*)persist_probe<2, 0u, 1, false>, dim3(grid), dim3(BLOCK), args, 0, 0));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned err = 0; CK(hipMemcpy(&err, pc.err, 4, hipMemcpyDeviceToHost));
                if (rep == 2) printf("n=%llu grid=%d: 1 persistent launch f=%d: %8.1f us per step (err=%u)\n", (unsigned long long)b.n, grid, fences, ms * 1000 / STEPS, err);
            }
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == -4) {  // do the three streams of the step kernel collide in DRAM banks? stagger their bases
        const size_t pad = 64u << 20;
        double *big_u, *big_v;
        CK(hipMalloc(&big_u, bytes + pad));
        CK(hipMalloc(&big_v, bytes + pad));
        const size_t offs[] = {0, 256, 4096, 8192, 65536, 65536 + 4096, 1u << 20, (1u << 20) + 8192, (2u << 20) + 4096, (16u << 20) + 65536 + 4096};
        printf("bases: q=%p u=%p v=%p\n", (void*)b.q, (void*)big_u, (void*)big_v);
        for (size_t o : offs) {
            Bufs c = b;
            c.u = reinterpret_cast<double*>(reinterpret_cast<char*>(big_u) + o);
            c.v = reinterpret_cast<double*>(reinterpret_cast<char*>(big_v) + 2 * o);
            CK(hipMemcpy(c.u, h.data(), bytes, hipMemcpyHostToDevice));
            CK(hipMemcpy(c.v, h.data(), bytes, hipMemcpyHostToDevice));
            printf("u += %zu B, v += %zu B: ", o, 2 * o);
            sweep_step<2, ~0u, ~0u, 1, 1>(c, {216});
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == -3) {  // two steps of the recursion as peek (3r) + pair (4r 1w) against 2 x (3r 1w)
        double* extra[4];
        for (auto& e : extra) { CK(hipMalloc(&e, bytes)); CK(hipMemcpy(e, h.data(), bytes, hipMemcpyHostToDevice)); }
        std::vector<int> g1 = {192, 216, 240, 256, 320, 432, 512};
        constexpr unsigned ALL = ~0u;
        sweep_step<2, ALL, ALL, 1, 1>(b, {216});
        sweep_peek_pair<1, ALL, ALL, 1, 1>(b, g1, extra);
        sweep_peek_pair<2, ALL, ALL, 1, 1>(b, g1, extra);
        sweep_peek_pair<4, ALL, ALL, 1, 1>(b, g1, extra);
        sweep_peek_pair<2, ALL, ALL, 2, 1>(b, g1, extra);
        sweep_peek_pair<4, ALL, ALL, 2, 1>(b, g1, extra);
        sweep_peek_pair<2, ALL, ALL, 2, 2>(b, g1, extra);
        sweep_peek_pair<2, ALL, ALL, 0, 1>(b, g1, extra);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == -1) {  // fixed-overhead probe: tiny vectors, back-to-back launches
        std::vector<int> g1 = {216};
        constexpr unsigned NO = 0u;
        sweep_copy<4, NO, NO, 2>(b, g1);
        sweep_dot<4, NO, 2>(b, g1);
        sweep_step<2, NO, NO, 1, 1>(b, g1);
        // 20 dependent step kernels back to back, one event pair around all of them
        OpTwoLoopStep<false, false, 0> op{};
        op.in[0] = b.q; op.in[1] = b.u; op.in[2] = b.v; op.out[0] = b.q;
        op.dot_in = b.board; op.ys_j = b.board + 1; op.alpha_j = b.board + 10; op.gamma_num = b.board; op.gamma_den = b.board + 1;
        op.mode_b = 1;
        RedCtl red{};
        red.partials = b.partials; red.ticket = b.ticket;
        for (int k = 0; k < RED_PTRS; ++k) red.out[k] = b.board + 2 + k;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) {
                handoff(red, b);
                hipLaunchKernelGGL((stream_kernel<decltype(op), 2, 0u, 0u, 1, 1>), dim3(216), dim3(BLOCK), 0, 0, op, b.n, 0, red);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("20 dependent step kernels at n=%llu: %.1f us each\n", (unsigned long long)b.n, ms * 1000 / 20);
        }
        {   // the same chain without a reduction (axpy with a device coefficient): launch + ramp only
            OpAxpy ax{};
            ax.in[0] = b.q; ax.in[1] = b.u; ax.out[0] = b.q; ax.c_host = 0.0; ax.c_dev = b.board;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; ++i)
                    hipLaunchKernelGGL((stream_kernel<OpAxpy, 2, 0u, 0u, 1, 1>), dim3(216), dim3(BLOCK), 0, 0, ax, b.n, 0, red);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("20 dependent axpy kernels (no reduction) at n=%llu: %.1f us each\n", (unsigned long long)b.n, ms * 1000 / 20);
            }
        }
        return 0;
    }
    // 22 vectors for the Gram kernels
    double* vecs[22];
    CK(hipMalloc(&b.board, 256 * sizeof(double)));
    { double bd2[256]; for (int i = 0; i < 256; ++i) bd2[i] = 1e-3; CK(hipMemcpy(b.board, bd2, sizeof(bd2), hipMemcpyHostToDevice)); }
    for (int i = 0; i < 22; ++i) { CK(hipMalloc(&vecs[i], bytes)); CK(hipMemcpy(vecs[i], h.data(), bytes, hipMemcpyHostToDevice)); }
    std::vector<int> grids = {128, 160, 192, 216, 240, 256, 432, 512, 864};
    if (argc > 2) { grids.clear(); for (int i = 2; i < argc; ++i) grids.push_back(atoi(argv[i])); }
    constexpr unsigned ALL = ~0u;
    sweep_rows<1>(b, grids, vecs);
    sweep_rows<2>(b, grids, vecs);
    sweep_rows<3>(b, grids, vecs);
    sweep_combine<1, ALL, ALL, 1, 1>(b, grids, vecs);
    sweep_combine<1, ALL, ALL, 2, 2>(b, grids, vecs);
    sweep_combine<1, ALL, ALL, 2, 4>(b, grids, vecs);
    sweep_combine<2, ALL, ALL, 1, 1>(b, grids, vecs);
    sweep_combine<2, ALL, ALL, 2, 1>(b, grids, vecs);
    return 0;
}
