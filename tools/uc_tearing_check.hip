// tools/uc_tearing_check.hip -- are aligned 8-byte agent-scope stores / loads single-copy atomic on UNCACHED device memory?
// One writer workgroup keeps storing words whose two halves are equal (i << 32 | i); 255 reader workgroups load them and
// count words whose halves differ (a torn store or a torn load) and words that go BACKWARDS (an older value after a newer).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/uc_tearing_check.hip -o tools/bin/uc_tearing_check
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned long long* w, int nwords, unsigned int iters, unsigned long long* bad, unsigned int* stop) {
    if (blockIdx.x == 0) {
        if (threadIdx.x < (unsigned)nwords) {
            for (unsigned int i = 1; i <= iters; ++i)
                __hip_atomic_store(w + threadIdx.x, ((unsigned long long)i << 32) | i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    unsigned long long torn = 0, back = 0, reads = 0;
    unsigned int last = 0;
    const int slot = threadIdx.x % nwords;
    for (;;) {
        const unsigned long long v = __hip_atomic_load(w + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
        torn += lo != hi;
        back += hi < last;
        last = hi > last ? hi : last;
        ++reads;
        if ((reads & 255) == 0 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (torn) atomicAdd(bad, torn);
    if (back) atomicAdd(bad + 1, back);
    atomicAdd(bad + 2, reads);
}
int main() {
    for (int uc = 1; uc >= 0; --uc) {
        void* w = nullptr;
        if (uc) hipExtMallocWithFlags(&w, 4096, hipDeviceMallocUncached); else hipMalloc(&w, 4096);
        hipMemset(w, 0, 4096);
        unsigned long long* bad; unsigned int* stop;
        hipMalloc(&bad, 64); hipMalloc(&stop, 64);
        hipMemset(bad, 0, 64); hipMemset(stop, 0, 64);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, (unsigned long long*)w, 32, 2000000u, bad, stop);
        hipDeviceSynchronize();
        unsigned long long h[3];
        hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost);
        printf("%s memory: %llu loads, %llu with unequal halves (torn), %llu going backwards\n", uc ? "uncached" : "plain   ", h[2], h[0], h[1]);
        hipFree(w); hipFree(bad); hipFree(stop);
    }
    return 0;
}
