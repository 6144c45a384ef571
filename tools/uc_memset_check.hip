// tools/uc_memset_check.hip -- does hipMemsetAsync clear recycled UNCACHED device memory as seen by agent-scope atomic loads?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/uc_memset_check.hip -o tools/bin/uc_memset_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void fill(unsigned long long* p, size_t n, unsigned long long v) {
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_store(p + i, v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void count(const unsigned long long* p, size_t n, unsigned long long* nz) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        c += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
    if (c) atomicAdd(nz, c);
}
int main() {
    const size_t n = (size_t)64 * 4096 * 2;  // the library's granule buffer: 4 MiB
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* nz;
    hipMalloc(&nz, 8);
    for (int round = 0; round < 6; ++round) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, n * 8, hipDeviceMallocUncached) != hipSuccess) { printf("alloc failed\n"); return 1; }
        unsigned long long before = 0, after = 0, after_copy = 0;
        hipMemsetAsync(nz, 0, 8, s);
        hipLaunchKernelGGL(count, dim3(256), dim3(256), 0, s, (unsigned long long*)p, n, nz);
        hipMemcpyAsync(&before, nz, 8, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        hipMemsetAsync(p, 0, n * 8, s);
        hipStreamSynchronize(s);
        hipMemsetAsync(nz, 0, 8, s);
        hipLaunchKernelGGL(count, dim3(256), dim3(256), 0, s, (unsigned long long*)p, n, nz);
        hipMemcpyAsync(&after, nz, 8, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        std::vector<unsigned long long> h(n);
        hipMemcpy(h.data(), p, n * 8, hipMemcpyDeviceToHost);
        for (auto v : h) after_copy += v != 0;
        printf("round %d: %p  non-zero words as allocated %llu, after hipMemsetAsync+sync %llu (kernel, sc1 loads) / %llu (hipMemcpy)\n", round, p,
               before, after, after_copy);
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, s, (unsigned long long*)p, n, 0x100000001ull);
        hipStreamSynchronize(s);
        hipFree(p);
    }
    return 0;
}
