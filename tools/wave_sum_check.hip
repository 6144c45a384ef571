// tools/wave_sum_check.hip -- does stream.h's wave_sum_dpp add what its comment says, in the order it says?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Irust-lbfgs_amd/csrc tools/wave_sum_check.hip -o tools/bin/wave_sum_check
// Random doubles per lane; the host forms the documented order (rows of 16 as trees with strides 8,4,2,1, then
// ((r0+r1)+r2)+r3) and wave_sum's order (tree with strides 32..1) and compares BITS; also times both on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "stream.h"
using namespace lh;

__global__ void k_check(const double* in, double* out_dpp, double* out_shfl, long long* cyc, int reps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = in[i];
    const double a = wave_sum_dpp(v);
    const double b = wave_sum(v);
    out_dpp[i] = a;
    out_shfl[i] = b;
    // timing: dependent chains of reductions
    double x = v;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) x = wave_sum_dpp(x) * 0.5 + v;
    long long t1 = clock64();
    double y = v;
    for (int r = 0; r < reps; ++r) y = __shfl(wave_sum(y), 0, 64) * 0.5 + v;
    long long t2 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
    if (x + y == 12345.678) out_dpp[i] = 0;  // keep the chains alive
}

static double tree16(const double* v) {  // strides 8, 4, 2, 1 within one row, as lane 0 sees it
    double t[16];
    memcpy(t, v, sizeof(t));
    for (int k = 8; k >= 1; k >>= 1)
        for (int i = 0; i < 16; ++i) t[i] = t[i] + (i + k < 16 ? t[i + k] : 0.0);
    return t[0];
}
static double tree64(const double* v) {  // wave_sum: strides 32 .. 1
    double t[64];
    memcpy(t, v, sizeof(t));
    for (int k = 32; k >= 1; k >>= 1)
        for (int i = 0; i + k < 64; ++i) t[i] = t[i] + t[i + k];
    return t[0];
}

int main() {
    const int waves = 64, n = waves * 64, reps = 2000;
    std::vector<double> h(n), a(n), b(n);
    srand(7);
    for (auto& x : h) x = (rand() / (double)RAND_MAX - 0.5) * exp2((double)(rand() % 40 - 20));
    double *din, *da, *db;
    long long* dc;
    hipMalloc(&din, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, 16);
    hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(waves / 4), dim3(256), 0, 0, din, da, db, dc, reps);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    long long cyc[2];
    hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
    int bad_dpp = 0, bad_uniform = 0, bad_shfl = 0;
    for (int w = 0; w < waves; ++w) {
        const double* v = h.data() + w * 64;
        const double want = ((tree16(v) + tree16(v + 16)) + tree16(v + 32)) + tree16(v + 48);
        for (int l = 0; l < 64; ++l)
            if (memcmp(&a[w * 64 + l], &want, 8) != 0) { if (l == 0) ++bad_dpp; else ++bad_uniform; }
        const double want2 = tree64(v);
        if (memcmp(&b[w * 64], &want2, 8) != 0) ++bad_shfl;
    }
    printf("wave_sum_dpp: %d of %d waves wrong in lane 0, %d other lanes differ (must be uniform); wave_sum: %d wrong\n", bad_dpp, waves,
           bad_uniform, bad_shfl);
    printf("cycles per reduction (dependent chain): dpp %.1f, bpermute tree (+broadcast) %.1f\n", cyc[0] / (double)reps, cyc[1] / (double)reps);
    return (bad_dpp || bad_uniform || bad_shfl) ? 1 : 0;
}
