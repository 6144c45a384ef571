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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bufs {
    double *q, *u, *v, *board, *partials;
    unsigned* ticket;
    uint64_t n;
};

template <class Op, int UNR, unsigned NTL, unsigned NTS, int MAP, int SPAN = 1>
float run(const Op& op, const Bufs& b, int grid, int reps, int* occ) {
    RedCtl red{};
    red.partials = b.partials;
    red.ticket = b.ticket;
    for (int k = 0; k < RED_PTRS; ++k) red.out[k] = b.board + 2 + k;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, stream_kernel<Op, UNR, NTL, NTS, MAP, SPAN>, BLOCK, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<Op, UNR, NTL, NTS, MAP, SPAN>), dim3(grid), dim3(BLOCK), 0, 0, op, b.n, 0, red);
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
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
            for (int i = 0; i < 20; ++i)
                hipLaunchKernelGGL((stream_kernel<decltype(op), 2, 0u, 0u, 1, 1>), dim3(216), dim3(BLOCK), 0, 0, op, b.n, 0, red);
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
