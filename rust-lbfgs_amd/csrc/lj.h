// rust-lbfgs_amd/csrc/lj.h -- device-resident Lennard-Jones objectives (BASELINE config 5).
//
// The USER objective of the reference's examples/lj.rs (epsilon = sigma = 1): E = sum_{i<j} 4 (r^-12 - r^-6),
// gx = dE/dx (examples/lj.rs:20-64, :113-118).  Not part of the optimiser's hot path -- it is the `evaluate`
// closure kept in HBM so that x and g never cross PCIe.  Two forms:
//   all-pairs  exact semantics of the example, O(N^2): one thread per atom i, positions of the atoms j staged through
//              LDS in tiles of 256, every pair evaluated from both ends (no atomics => deterministic); the j range is
//              split over several workgroups when the tile count alone would not fill the chip evenly;
//   neighbours a fixed ELL neighbour table (max_nbr x natoms, column-major, -1 = empty) and a cutoff:
//              E = sum_{i<j, r<rc} [v(r) - v(rc)], the substitute evaluator for 1e6 atoms (documented deviation).
#pragma once
#include "stream.h"

namespace lh {

__device__ __forceinline__ void lj_pair(double dx, double dy, double dz, double& e, double& fx, double& fy, double& fz) {
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double inv2 = 1.0 / r2;
    const double s6 = inv2 * inv2 * inv2;
    e += 4.0 * (s6 * s6 - s6);                       // pair_energy   lj.rs:22-25
    const double c = 24.0 * (s6 - 2.0 * (s6 * s6)) * inv2;  // pair_gradient/r  lj.rs:28-32, :57-58
    fx += c * dx; fy += c * dy; fz += c * dz;        // gx_i += g(r)/r * (x_i - x_j)
}

// Work unit = (tile of BLOCK atoms i) x (one of S contiguous ranges of atoms j).  S > 1 spreads a system whose tile
// count does not fill or evenly divide the chip over more, shorter workgroups; each range's forces go to their own
// slice of `out` (S x 3N doubles) and lj_combine_kernel adds the slices in order (no atomics => deterministic).
__global__ __launch_bounds__(BLOCK) void lj_allpairs_kernel(const double* __restrict__ x, double* __restrict__ out,
                                                             const uint32_t natoms, const uint32_t ntiles,
                                                             const uint32_t jspan, const RedCtl red) {
    __shared__ double tile[3 * BLOCK];
    const uint32_t it = blockIdx.x % ntiles, js = blockIdx.x / ntiles;
    const uint32_t i = it * BLOCK + threadIdx.x;
    const bool live = i < natoms;
    double xi = 0, yi = 0, zi = 0;
    if (live) { xi = x[3 * (size_t)i]; yi = x[3 * (size_t)i + 1]; zi = x[3 * (size_t)i + 2]; }
    double e = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
    const uint32_t j_lo = js * jspan, j_hi = min(natoms, j_lo + jspan);  // jspan is a multiple of BLOCK
    for (uint32_t base = j_lo; base < j_hi; base += BLOCK) {
        const uint32_t cnt = min((uint32_t)BLOCK, j_hi - base);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < 3 * cnt; t += BLOCK) tile[t] = x[3 * (size_t)base + t];
        __syncthreads();
        if (live) {
            if (base != it * BLOCK) {
                // a tile that cannot contain atom i: no self test, so the loop unrolls and four independent
                // division chains keep the f64 pipe busy
#pragma unroll 4
                for (uint32_t j = 0; j < cnt; ++j)
                    lj_pair(xi - tile[3 * j], yi - tile[3 * j + 1], zi - tile[3 * j + 2], e, fx, fy, fz);
            } else {
                for (uint32_t j = 0; j < cnt; ++j) {
                    if (base + j == i) continue;
                    lj_pair(xi - tile[3 * j], yi - tile[3 * j + 1], zi - tile[3 * j + 2], e, fx, fy, fz);
                }
            }
        }
    }
    if (live) {
        double* g = out + (size_t)js * 3 * natoms;
        g[3 * (size_t)i] = fx; g[3 * (size_t)i + 1] = fy; g[3 * (size_t)i + 2] = fz;
    }
    double acc[1] = {0.5 * e};  // every pair was seen from both ends
    grid_reduce<1>(acc, red);
}

// g = slice_0 + slice_1 + ... + slice_{S-1}, in that order
__global__ __launch_bounds__(BLOCK) void lj_combine_kernel(const double* __restrict__ slices, double* __restrict__ g,
                                                            const uint64_t n, const uint32_t nslices) {
    for (uint64_t k = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; k < n; k += (uint64_t)gridDim.x * BLOCK) {
        double t = slices[k];
        for (uint32_t s = 1; s < nslices; ++s) t += slices[(size_t)s * n + k];
        g[k] = t;
    }
}

__global__ __launch_bounds__(BLOCK) void lj_neighbors_kernel(const double* __restrict__ x, double* __restrict__ g,
                                                              const int32_t* __restrict__ nbr, const uint32_t max_nbr,
                                                              const uint32_t natoms, const double rc2, const double eshift,
                                                              const RedCtl red) {
    double e = 0.0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < natoms; i += gridDim.x * BLOCK) {
        const double xi = x[3 * (size_t)i], yi = x[3 * (size_t)i + 1], zi = x[3 * (size_t)i + 2];
        double fx = 0.0, fy = 0.0, fz = 0.0;
        // Branch-free and unrolled: the gathers of four neighbours are in flight together.  An empty slot (-1) or a pair
        // beyond the cutoff is evaluated at a harmless r^2 = 1 and adds a selected 0.0, so the sums are those of the
        // branching loop.
#pragma unroll 4
        for (uint32_t k = 0; k < max_nbr; ++k) {
            const int32_t j = nbr[(size_t)k * natoms + i];  // coalesced across lanes
            const size_t jj = (j < 0) ? (size_t)i : (size_t)j;
            const double dx = xi - x[3 * jj], dy = yi - x[3 * jj + 1], dz = zi - x[3 * jj + 2];
            const double r2 = dx * dx + dy * dy + dz * dz;
            const bool on = (j >= 0) && (r2 < rc2);
            const double inv2 = 1.0 / (on ? r2 : 1.0);
            const double s6 = inv2 * inv2 * inv2;
            e += on ? 4.0 * (s6 * s6 - s6) : 0.0;                     // pair_energy   lj.rs:22-25
            const double c = 24.0 * (s6 - 2.0 * (s6 * s6)) * inv2;   // pair_gradient/r  lj.rs:28-32, :57-58
            fx += on ? c * dx : 0.0; fy += on ? c * dy : 0.0; fz += on ? c * dz : 0.0;  // selects: a NaN neighbour stays out
            e -= on ? eshift : 0.0;
        }
        g[3 * (size_t)i] = fx; g[3 * (size_t)i + 1] = fy; g[3 * (size_t)i + 2] = fz;
    }
    double acc[1] = {0.5 * e};
    grid_reduce<1>(acc, red);
}

}  // namespace lh
