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

// The LJ evaluators are the USER objective (examples/lj.rs), not an L-BFGS primitive: their parity bar is 1e-10
// relative against the oracle, not the bit-exactness of the math.rs operators.  So, unlike the rest of the translation
// unit (-ffp-contract=off), the pair arithmetic may contract mul+add into FMA, and 1/r^2 is a hardware reciprocal
// estimate (v_rcp_f64, ~2^-23) polished by two Newton steps (4 FMAs, <= 1 ulp) instead of the ~10-instruction IEEE
// division sequence.  The pragma is switched off again at the end of this header.
#pragma clang fp contract(fast)

__device__ __forceinline__ double lj_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    return r;
}

__device__ __forceinline__ void lj_pair(double dx, double dy, double dz, double& e, double& fx, double& fy, double& fz) {
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double inv2 = lj_rcp(r2);
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
    const DevCounters c0 = load_counters(red);
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
    grid_reduce<1>(acc, red, c0);
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
    const DevCounters c0 = load_counters(red);
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
            const double inv2 = lj_rcp(on ? r2 : 1.0);
            const double s6 = inv2 * inv2 * inv2;
            e += on ? 4.0 * (s6 * s6 - s6) : 0.0;                     // pair_energy   lj.rs:22-25
            const double c = 24.0 * (s6 - 2.0 * (s6 * s6)) * inv2;   // pair_gradient/r  lj.rs:28-32, :57-58
            fx += on ? c * dx : 0.0; fy += on ? c * dy : 0.0; fz += on ? c * dz : 0.0;  // selects: a NaN neighbour stays out
            e -= on ? eshift : 0.0;
        }
        g[3 * (size_t)i] = fx; g[3 * (size_t)i + 1] = fy; g[3 * (size_t)i + 2] = fz;
    }
    double acc[1] = {0.5 * e};
    grid_reduce<1>(acc, red, c0);
}

// ---------------------------------------------------------------------------------------------------------------
// Rebuildable neighbour structure for the cutoff form (LBFGS_HIP_OBJ_LJ_CELLS): a Verlet list with a skin, built on
// the device from a cell list.
//   list radius rl = rc + skin; the list stays complete for the cutoff rc while no atom has moved more than skin/2
//   from where it was at build time (xref).  The evaluation kernel counts the atoms beyond that bound in the same
//   pass; the host rebuilds and re-evaluates when the count is non-zero, so every result is the exact cutoff sum
//   E = sum_{i<j, r<rc} [v(r) - v(rc)] whatever the rebuild history.
//   build: bounding box -> cells of side >= rl -> count (atomics) -> exclusive scan -> fill (atomics) -> each cell's
//   atoms sorted by index -> per atom the 27 surrounding cells in a fixed order -> ELL table (column-major, padded to a
//   multiple of 4 with -1).  The sort makes the table, and with it the order of every force sum, independent of the
//   order in which the atomics landed: results are bitwise reproducible.
struct LjGrid {
    double ox, oy, oz, inv_side;
    int nx, ny, nz;
};

__device__ __forceinline__ int lj_cell_coord(double v, double o, double inv_side, int nmax) {
    const double t = (v - o) * inv_side;
    int c = (t >= 0.0) ? (int)fmin(t, 2.0e9) : 0;  // NaN compares false -> 0
    return c < nmax ? c : nmax - 1;
}

// per-workgroup min / max of the coordinates: part[6*blockIdx.x + {0,1,2}] = min, {3,4,5} = max (the host reduces them)
__global__ __launch_bounds__(BLOCK) void lj_bbox_kernel(const double* __restrict__ x, const uint32_t natoms,
                                                         double* __restrict__ part) {
    __shared__ double sh[6][WAVES];
    double lo[3] = {1.0 / 0.0, 1.0 / 0.0, 1.0 / 0.0}, hi[3] = {-1.0 / 0.0, -1.0 / 0.0, -1.0 / 0.0};
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < natoms; i += gridDim.x * BLOCK)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = x[3 * (size_t)i + k];
            lo[k] = fmin(lo[k], v);
            hi[k] = fmax(hi[k], v);
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fmin(lo[k], __shfl_down(lo[k], off, 64));
            hi[k] = fmax(hi[k], __shfl_down(hi[k], off, 64));
        }
        if (lane == 0) { sh[k][wave] = lo[k]; sh[3 + k][wave] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = sh[threadIdx.x][0];
        for (int w = 1; w < WAVES; ++w) v = (threadIdx.x < 3) ? fmin(v, sh[threadIdx.x][w]) : fmax(v, sh[threadIdx.x][w]);
        part[6 * (size_t)blockIdx.x + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(BLOCK) void lj_cells_count_kernel(const double* __restrict__ x, const uint32_t natoms,
                                                                const LjGrid gr, int32_t* __restrict__ cell_of,
                                                                int32_t* __restrict__ slot_of, int32_t* __restrict__ counts) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < natoms; i += gridDim.x * BLOCK) {
        const int cx = lj_cell_coord(x[3 * (size_t)i], gr.ox, gr.inv_side, gr.nx);
        const int cy = lj_cell_coord(x[3 * (size_t)i + 1], gr.oy, gr.inv_side, gr.ny);
        const int cz = lj_cell_coord(x[3 * (size_t)i + 2], gr.oz, gr.inv_side, gr.nz);
        const int c = (cz * gr.ny + cy) * gr.nx + cx;
        cell_of[i] = c;
        slot_of[i] = atomicAdd(&counts[c], 1);  // a place of its own within the cell (arbitrary order: the cells are sorted later)
    }
}

// exclusive scan of counts[0, ncells) into starts[0, ncells] (and cursor = starts), three small launches:
//   tiles of LJ_SCAN_TILE cells -> per-tile sums ; one workgroup scans the (<= 1024) tile sums ; every tile scans itself
// with its offset.  All loads coalesced; ~10 us at 64k cells where a single serial workgroup took 170 us.
constexpr int LJ_SCAN_TILE = 2048;   // 256 threads x 8 cells
__device__ __forceinline__ int32_t lj_block_exclusive_scan(int32_t v, int32_t* lds /* [WAVES] */, int32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const int32_t s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}
__global__ __launch_bounds__(BLOCK) void lj_scan_tile_sums_kernel(const int32_t* __restrict__ counts, const uint32_t ncells,
                                                                   int32_t* __restrict__ tile_sums) {
    __shared__ int32_t lds[WAVES];
    const uint32_t base = blockIdx.x * LJ_SCAN_TILE;
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < LJ_SCAN_TILE / BLOCK; ++k) {
        const uint32_t c = base + k * BLOCK + threadIdx.x;
        s += (c < ncells) ? counts[c] : 0;
    }
    int32_t total;
    (void)lj_block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void lj_scan_tile_offsets_kernel(int32_t* __restrict__ tile_sums, const uint32_t ntiles) {
    __shared__ int32_t sums[1024];
    const int32_t v = (threadIdx.x < ntiles) ? tile_sums[threadIdx.x] : 0;
    sums[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan over <= 1024 tile sums
        const int32_t t = (threadIdx.x >= (unsigned)off) ? sums[threadIdx.x - off] : 0;
        __syncthreads();
        sums[threadIdx.x] += t;
        __syncthreads();
    }
    if (threadIdx.x < ntiles) tile_sums[threadIdx.x] = sums[threadIdx.x] - v;  // exclusive
}
__global__ __launch_bounds__(BLOCK) void lj_scan_apply_kernel(const int32_t* __restrict__ counts, const uint32_t ncells,
                                                               const int32_t* __restrict__ tile_offsets,
                                                               int32_t* __restrict__ starts, int32_t* __restrict__ cursor) {
    __shared__ int32_t lds[WAVES];
    // thread t owns the 8 consecutive cells [base + 8t, base + 8t + 8)
    const uint32_t c0 = blockIdx.x * LJ_SCAN_TILE + threadIdx.x * (LJ_SCAN_TILE / BLOCK);
    int32_t v[LJ_SCAN_TILE / BLOCK], s = 0;
#pragma unroll
    for (int k = 0; k < LJ_SCAN_TILE / BLOCK; ++k) {
        v[k] = (c0 + k < ncells) ? counts[c0 + k] : 0;
        s += v[k];
    }
    int32_t total;
    int32_t run = tile_offsets[blockIdx.x] + lj_block_exclusive_scan(s, lds, total);
#pragma unroll
    for (int k = 0; k < LJ_SCAN_TILE / BLOCK; ++k) {
        if (c0 + k < ncells) { starts[c0 + k] = run; cursor[c0 + k] = run; }
        run += v[k];
        if (c0 + k + 1 == ncells) starts[ncells] = run;
    }
}

// (no second round of atomics: the count kernel's atomicAdd already handed every atom its place within its cell)
__global__ __launch_bounds__(BLOCK) void lj_cells_fill_kernel(const int32_t* __restrict__ cell_of, const int32_t* __restrict__ slot_of,
                                                               const uint32_t natoms, const int32_t* __restrict__ starts,
                                                               int32_t* __restrict__ sorted) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < natoms; i += gridDim.x * BLOCK)
        sorted[starts[cell_of[i]] + slot_of[i]] = (int32_t)i;
}

// one thread per cell: its atoms into ascending index order (insertion sort: a cell holds a few dozen atoms).  Cells of up
// to LJ_SORT_LDS atoms are sorted in an LDS column of their thread (buf[k][tid]: conflict-free) instead of in place in
// global memory, where every compare-and-shift was a dependent load and store (72 -> 25 us at 1e6 atoms).
constexpr int LJ_SORT_LDS = 32;
__global__ __launch_bounds__(BLOCK) void lj_cells_sort_kernel(const int32_t* __restrict__ starts, const uint32_t ncells,
                                                               int32_t* __restrict__ sorted) {
    __shared__ int32_t buf[LJ_SORT_LDS][BLOCK];
    const uint32_t tid = threadIdx.x;
    for (uint32_t c = blockIdx.x * BLOCK + tid; c < ncells; c += gridDim.x * BLOCK) {
        const int32_t a = starts[c], b = starts[c + 1], len = b - a;
        if (len <= LJ_SORT_LDS) {
            for (int32_t t = 0; t < len; ++t) buf[t][tid] = sorted[a + t];
            for (int32_t t = 1; t < len; ++t) {
                const int32_t v = buf[t][tid];
                int32_t u = t - 1;
                while (u >= 0 && buf[u][tid] > v) { buf[u + 1][tid] = buf[u][tid]; --u; }
                buf[u + 1][tid] = v;
            }
            for (int32_t t = 0; t < len; ++t) sorted[a + t] = buf[t][tid];
        } else {
            for (int32_t t = a + 1; t < b; ++t) {
                const int32_t v = sorted[t];
                int32_t u = t - 1;
                while (u >= a && sorted[u] > v) { sorted[u + 1] = sorted[u]; --u; }
                sorted[u + 1] = v;
            }
        }
    }
}

// positions in cell order, one 32-byte record per atom: xs[4t .. 4t+2] = x[sorted[t]], and the low word of xs[4t+3] =
// sorted[t] (the list kernel reads its candidates sequentially: two aligned 16-byte loads bring position AND index)
__global__ __launch_bounds__(BLOCK) void lj_cells_gather_kernel(const double* __restrict__ x, const uint32_t natoms,
                                                                 const int32_t* __restrict__ sorted, double* __restrict__ xs) {
    for (uint32_t t = blockIdx.x * BLOCK + threadIdx.x; t < natoms; t += gridDim.x * BLOCK) {
        const size_t j = (size_t)sorted[t];
        d2* rec = reinterpret_cast<d2*>(xs + 4 * (size_t)t);
        rec[0] = d2{x[3 * j], x[3 * j + 1]};
        rec[1] = d2{x[3 * j + 2], __hiloint2double(0, (int)j)};
    }
}

#ifndef LH_LJ_EVAL_GROUP
#define LH_LJ_EVAL_GROUP 16
#endif
constexpr uint32_t LJ_EVAL_GROUP = LH_LJ_EVAL_GROUP;  // list entries the evaluation takes at a time (a multiple of 4; 16 measured best of 4 ... 32: more gathers in flight): lists are padded to it with -1
// close an atom's list of k entries (k <= max_nbr, a multiple of LJ_EVAL_GROUP): flush the partly filled group of four, pad
// with -1 up to the next multiple of LJ_EVAL_GROUP; -> the padded length
__device__ __forceinline__ uint32_t lj_list_finish(int32_t* __restrict__ mine, const uint32_t k, const int4 q4) {
    uint32_t kk = (k + 3u) & ~3u;
    if (k & 3u) *reinterpret_cast<int4*>(mine + (k & ~3u)) = q4;  // (its empty slots hold -1 already)
    for (; kk % LJ_EVAL_GROUP != 0u; kk += 4u) *reinterpret_cast<int4*>(mine + kk) = make_int4(-1, -1, -1, -1);
    return kk;
}
#ifndef LH_LJ_BUILD_UNROLL
#define LH_LJ_BUILD_UNROLL 8
#endif
constexpr int LJ_BUILD_UNROLL = LH_LJ_BUILD_UNROLL;
// entry number k of an atom's list: four entries are collected in registers and leave as ONE 16-byte store (a 4-byte
// store per hit costs the memory pipeline a cache-line access per lane and hit: 0.4 ms of the kernel at 1e6 atoms)
__device__ __forceinline__ void lj_list_push(int32_t* __restrict__ mine, const uint32_t max_nbr, uint32_t& k, int4& q4, const int32_t v) {
    const uint32_t slot = k & 3u;
    q4.x = slot == 0u ? v : q4.x;
    q4.y = slot == 1u ? v : q4.y;
    q4.z = slot == 2u ? v : q4.z;
    q4.w = slot == 3u ? v : q4.w;
    if (slot == 3u) {
        if (k < max_nbr) *reinterpret_cast<int4*>(mine + (k - 3u)) = q4;  // (max_nbr is a multiple of 4: the group fits or not as a whole)
        q4 = make_int4(-1, -1, -1, -1);
    }
    ++k;
}
// Verlet list of atom i = sorted[t]: every j != i with |x_i - x_j| < rl, in the order (27 cells: z, y, x ascending; atoms
// of a cell ascending).  One thread per atom IN CELL ORDER: the threads of a wave sit in the same few cells and walk the
// same contiguous candidate segments of xs (the three cells of a row are adjacent in memory).
// cnt[i] = entries used (a multiple of 4, the tail padded with -1); *overflow = the largest list that did not fit.
// The list is written ROW-major here (nbr[i * max_nbr + k]: a thread's hits go to consecutive addresses) and turned into
// the column-major table the evaluation reads coalesced by lj_cells_transpose_kernel: a thread finds its k-th neighbour
// at a data-dependent moment, and column-major stores from here put every 4-byte entry into a cache line of its own --
// 0.7 of this kernel's 1.13 ms at 1e6 atoms went into those (measured by compiling the stores out).
__global__ __launch_bounds__(BLOCK) void lj_cells_build_kernel(const double* __restrict__ xs, const uint32_t natoms,
                                                                const LjGrid gr, const int32_t* __restrict__ cell_of,
                                                                const int32_t* __restrict__ starts,
                                                                const int32_t* __restrict__ sorted, const double rl2,
                                                                const uint32_t max_nbr, int32_t* __restrict__ nbr,
                                                                int32_t* __restrict__ cnt, double* __restrict__ xref,
                                                                unsigned int* __restrict__ overflow) {
    for (uint32_t t = blockIdx.x * BLOCK + threadIdx.x; t < natoms; t += gridDim.x * BLOCK) {
        const uint32_t i = (uint32_t)sorted[t];
        const double xi = xs[4 * (size_t)t], yi = xs[4 * (size_t)t + 1], zi = xs[4 * (size_t)t + 2];
        xref[3 * (size_t)i] = xi; xref[3 * (size_t)i + 1] = yi; xref[3 * (size_t)i + 2] = zi;
        int32_t* __restrict__ mine = nbr + (size_t)i * max_nbr;
        int4 q4 = make_int4(-1, -1, -1, -1);  // the group of four entries being filled (stored 16 bytes at a time)  // this atom's list, contiguous (lj_cells_transpose_kernel turns the table)
        const int c = cell_of[i];
        const int cx = c % gr.nx, cy = (c / gr.nx) % gr.ny, cz = c / (gr.nx * gr.ny);
        const int ax0 = max(cx - 1, 0), ax1 = min(cx + 1, gr.nx - 1);
        uint32_t k = 0;
        for (int dz = -1; dz <= 1; ++dz) {
            const int az = cz + dz;
            if (az < 0 || az >= gr.nz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int ay = cy + dy;
                if (ay < 0 || ay >= gr.ny) continue;
                const int row = (az * gr.ny + ay) * gr.nx;
                const int32_t u1 = starts[row + ax1 + 1];
                int32_t u = starts[row + ax0];
                // LJ_BUILD_UNROLL candidates at a time: their positions are loaded together and tested in order
                for (; u + LJ_BUILD_UNROLL <= u1; u += LJ_BUILD_UNROLL) {
                    double px[LJ_BUILD_UNROLL], py[LJ_BUILD_UNROLL], pz[LJ_BUILD_UNROLL];
                    int32_t sj[LJ_BUILD_UNROLL];  // (the candidates' atom indices travel with their positions: no load behind a hit)
#pragma unroll
                    for (int w = 0; w < LJ_BUILD_UNROLL; ++w) {
                        const d2* rec = reinterpret_cast<const d2*>(xs + 4 * (size_t)(u + w));
                        const d2 r0 = rec[0], r1 = rec[1];
                        px[w] = r0.x; py[w] = r0.y; pz[w] = r1.x;
                        sj[w] = __double2loint(r1.y);
                    }
#pragma unroll
                    for (int w = 0; w < LJ_BUILD_UNROLL; ++w) {
                        const double dx = xi - px[w], dyy = yi - py[w], dzz = zi - pz[w];
                        if (dx * dx + dyy * dyy + dzz * dzz < rl2 && (uint32_t)(u + w) != t) {
                            lj_list_push(mine, max_nbr, k, q4, sj[w]);
                        }
                    }
                }
                for (; u < u1; ++u) {
                    const double dx = xi - xs[4 * (size_t)u], dyy = yi - xs[4 * (size_t)u + 1], dzz = zi - xs[4 * (size_t)u + 2];
                    if (dx * dx + dyy * dyy + dzz * dzz < rl2 && (uint32_t)u != t) {
                        lj_list_push(mine, max_nbr, k, q4, sorted[u]);
                    }
                }
            }
        }
        if (k > max_nbr) {
            atomicMax(overflow, k);
            k = max_nbr;
        }
        cnt[i] = (int32_t)lj_list_finish(mine, k, q4);
    }
}

// ---- the same list from SINGLE-precision candidate tests (the default) ---------------------------------------------
// What the list has to be is a SUPERSET: every pair that can come within the cutoff before the next rebuild -- every
// pair closer than rl at build time -- must be in it; what else is in it does not matter, because the evaluation tests
// every entry against the cutoff in double precision and adds exact zeros for the rest (as it does for all the entries
// between rc and rl).  Entries keep their enumeration order either way, so every sum is formed in the same order from
// the same non-zero terms: the evaluation's results are bit-identical to those over the double-precision list (tested).
// So the candidate test runs in fp32 on positions relative to the grid origin, against rl^2 plus a margin that covers
// the rounding of those positions (host: lj_cells_rebuild); 16-byte records (x, y, z, atom index) instead of 32, half the
// loads, and single-precision arithmetic that the compiler packs two candidates at a time.
typedef float lj_f2 __attribute__((ext_vector_type(2)));
struct LjRec32 {
    float x, y, z;
    int32_t idx;
};
__global__ __launch_bounds__(BLOCK) void lj_cells_gather32_kernel(const double* __restrict__ x, const uint32_t natoms,
                                                                   const int32_t* __restrict__ sorted, const LjGrid gr,
                                                                   LjRec32* __restrict__ xs32) {
    for (uint32_t t = blockIdx.x * BLOCK + threadIdx.x; t < natoms; t += gridDim.x * BLOCK) {
        const size_t j = (size_t)sorted[t];
        LjRec32 r;
        r.x = (float)(x[3 * j] - gr.ox); r.y = (float)(x[3 * j + 1] - gr.oy); r.z = (float)(x[3 * j + 2] - gr.oz);
        r.idx = (int32_t)j;
        *reinterpret_cast<int4*>(xs32 + t) = *reinterpret_cast<const int4*>(&r);
    }
}
__global__ __launch_bounds__(BLOCK) void lj_cells_build32_kernel(const LjRec32* __restrict__ xs32, const uint32_t natoms,
                                                                  const LjGrid gr, const int32_t* __restrict__ cell_of,
                                                                  const int32_t* __restrict__ starts, const float rl2m,
                                                                  const uint32_t max_nbr, int32_t* __restrict__ nbr,
                                                                  int32_t* __restrict__ cnt, unsigned int* __restrict__ overflow) {
    static_assert(LJ_BUILD_UNROLL % 2 == 0, "candidates are tested two at a time");
    for (uint32_t t = blockIdx.x * BLOCK + threadIdx.x; t < natoms; t += gridDim.x * BLOCK) {
        LjRec32 me;
        *reinterpret_cast<int4*>(&me) = *reinterpret_cast<const int4*>(xs32 + t);
        const uint32_t i = (uint32_t)me.idx;
        int32_t* __restrict__ mine = nbr + (size_t)i * max_nbr;
        int4 q4 = make_int4(-1, -1, -1, -1);
        const int c = cell_of[i];
        const int cx = c % gr.nx, cy = (c / gr.nx) % gr.ny, cz = c / (gr.nx * gr.ny);
        const int ax0 = max(cx - 1, 0), ax1 = min(cx + 1, gr.nx - 1);
        const lj_f2 xi = {me.x, me.x}, yi = {me.y, me.y}, zi = {me.z, me.z};
        uint32_t k = 0;
        for (int dz = -1; dz <= 1; ++dz) {
            const int az = cz + dz;
            if (az < 0 || az >= gr.nz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int ay = cy + dy;
                if (ay < 0 || ay >= gr.ny) continue;
                const int row = (az * gr.ny + ay) * gr.nx;
                const int32_t u1 = starts[row + ax1 + 1];
                int32_t u = starts[row + ax0];
                for (; u + LJ_BUILD_UNROLL <= u1; u += LJ_BUILD_UNROLL) {
                    LjRec32 cnd[LJ_BUILD_UNROLL];
#pragma unroll
                    for (int w = 0; w < LJ_BUILD_UNROLL; ++w)
                        *reinterpret_cast<int4*>(&cnd[w]) = *reinterpret_cast<const int4*>(xs32 + u + w);
#pragma unroll
                    for (int w = 0; w < LJ_BUILD_UNROLL; w += 2) {
                        const lj_f2 dx = xi - lj_f2{cnd[w].x, cnd[w + 1].x}, dyy = yi - lj_f2{cnd[w].y, cnd[w + 1].y},
                                    dzz = zi - lj_f2{cnd[w].z, cnd[w + 1].z};
                        const lj_f2 r2 = dx * dx + dyy * dyy + dzz * dzz;
                        if (r2.x < rl2m && (uint32_t)(u + w) != t) lj_list_push(mine, max_nbr, k, q4, cnd[w].idx);
                        if (r2.y < rl2m && (uint32_t)(u + w + 1) != t) lj_list_push(mine, max_nbr, k, q4, cnd[w + 1].idx);
                    }
                }
                for (; u < u1; ++u) {
                    LjRec32 cd;
                    *reinterpret_cast<int4*>(&cd) = *reinterpret_cast<const int4*>(xs32 + u);
                    const float dx = me.x - cd.x, dyy = me.y - cd.y, dzz = me.z - cd.z;
                    if (dx * dx + dyy * dyy + dzz * dzz < rl2m && (uint32_t)u != t) lj_list_push(mine, max_nbr, k, q4, cd.idx);
                }
            }
        }
        if (k > max_nbr) {
            atomicMax(overflow, k);
            k = max_nbr;
        }
        cnt[i] = (int32_t)lj_list_finish(mine, k, q4);
    }
}

// rows[i * max_nbr + k] -> table[k * natoms + i] for k < cnt[i], in tiles of 64 atoms x 64 entries through LDS: reads
// run along an atom's list (256 contiguous bytes per wave), writes along the atoms (256 contiguous bytes per wave and k)
__global__ __launch_bounds__(BLOCK) void lj_cells_transpose_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ cnt,
                                                                    const uint32_t natoms, const uint32_t max_nbr,
                                                                    int32_t* __restrict__ table) {
    __shared__ int32_t tile[64][65];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ntiles = (natoms + 63) / 64;
    for (uint32_t tb = blockIdx.x; tb < ntiles; tb += gridDim.x) {
        const uint32_t i0 = tb * 64;
        // the longest list of the tile (lists are padded to a multiple of 4 with -1 and read up to cnt only)
        uint32_t kmax = (i0 + lane < natoms) ? (uint32_t)cnt[i0 + lane] : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
        for (uint32_t k0 = 0; k0 < kmax; k0 += 64) {
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) {  // 16 lanes x 16 bytes = 64 consecutive entries of one atom's list; 16 atoms per pass
                const uint32_t idx = threadIdx.x + BLOCK * r, a = idx >> 4, c = idx & 15u;
                const uint32_t i = i0 + a, k = k0 + 4u * c;
                int4 v = make_int4(-1, -1, -1, -1);
                if (i < natoms && k < max_nbr) v = *reinterpret_cast<const int4*>(rows + (size_t)i * max_nbr + k);  // (max_nbr % 4 == 0)
                tile[a][4 * c] = v.x; tile[a][4 * c + 1] = v.y; tile[a][4 * c + 2] = v.z; tile[a][4 * c + 3] = v.w;
            }
            __syncthreads();
            for (uint32_t kk = wave; kk < 64; kk += WAVES) {  // entry k0 + kk of 64 consecutive atoms
                const uint32_t i = i0 + lane, k = k0 + kk;
                if (i < natoms && k < kmax) table[(size_t)k * natoms + i] = tile[lane][kk];
            }
        }
    }
}

// Has any atom left its skin/2 sphere since the list was built?  (*moved |= 1; NaN counts as moved.)  48 bytes per atom,
// run BEFORE every evaluation: an evaluation over a stale list is thrown away, and at 1e6 atoms it costs 0.41 ms
// against this kernel's ~15 us -- in a relaxing system every third evaluation was such a throw-away.
__global__ __launch_bounds__(BLOCK) void lj_cells_check_kernel(const double* __restrict__ x, const double* __restrict__ xref,
                                                                const uint32_t natoms, const double half_skin2,
                                                                unsigned int* __restrict__ moved) {
    bool any = false;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < natoms; i += gridDim.x * BLOCK) {
        const double ux = x[3 * (size_t)i] - xref[3 * (size_t)i], uy = x[3 * (size_t)i + 1] - xref[3 * (size_t)i + 1],
                     uz = x[3 * (size_t)i + 2] - xref[3 * (size_t)i + 2];
        any |= !(ux * ux + uy * uy + uz * uz <= half_skin2);
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(moved, 1u);
}

// take_line_step (core.rs:155-158: x = xp + t*d, a multiply then an add) fused with the staleness check above: the check reads
// every coordinate of x anyway, so a line-search trial of the LJ_CELLS objective forms its point here and saves a launch.
__global__ __launch_bounds__(BLOCK) void lj_cells_step_check_kernel(const double* __restrict__ xp, const double* __restrict__ d,
                                                                     const double t, double* __restrict__ x,
                                                                     const double* __restrict__ xref, const uint32_t natoms,
                                                                     const double half_skin2, unsigned int* __restrict__ moved) {
    bool any = false;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < natoms; i += gridDim.x * BLOCK) {
        const size_t b = 3 * (size_t)i;
        // (this header contracts a*b+c into FMAs for the pair arithmetic; the line step is the reference's multiply-then-add)
        const double x0 = __dadd_rn(xp[b], __dmul_rn(t, d[b])), x1 = __dadd_rn(xp[b + 1], __dmul_rn(t, d[b + 1])),
                     x2 = __dadd_rn(xp[b + 2], __dmul_rn(t, d[b + 2]));
        x[b] = x0; x[b + 1] = x1; x[b + 2] = x2;
        const double ux = x0 - xref[b], uy = x1 - xref[b + 1], uz = x2 - xref[b + 2];
        any |= !(ux * ux + uy * uy + uz * uz <= half_skin2);
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(moved, 1u);
}

// The cutoff sum over the Verlet list, and (second total) the number of atoms that have left their skin/2 sphere.
// DOT: a third total, g.d (dg_unchecked, core.rs:114-116) -- a trial's directional derivative summed where g is formed, so
// that the search needs no separate pass over g and d.
// `moved_flag` (lj_cells_check_kernel, same stream, just before): non-zero = the list is stale -- nothing is evaluated,
// the second total is reported as 1 and the host rebuilds; the reduction still runs (the launch's tags are spoken for).
template <bool ROWS, bool DOT = false>
__global__ __launch_bounds__(BLOCK) void lj_cells_eval_kernel(const double* __restrict__ x, double* __restrict__ g,
                                                               const int32_t* __restrict__ nbr,
                                                               const int32_t* __restrict__ cnt,
                                                               const double* __restrict__ xref, const uint32_t natoms,
                                                               const uint32_t max_nbr, const double rc2, const double eshift,
                                                               const double half_skin2,
                                                               const unsigned int* __restrict__ moved_flag, const RedCtl red,
                                                               const double* __restrict__ dir = nullptr) {
    const DevCounters c0 = load_counters(red);
    double e = 0.0, stale = 0.0, gd = 0.0;
    const bool skip = *moved_flag != 0u;
    if (skip) stale = (blockIdx.x == 0 && threadIdx.x == 0) ? 1.0 : 0.0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < (skip ? 0u : natoms); i += gridDim.x * BLOCK) {
        const double xi = x[3 * (size_t)i], yi = x[3 * (size_t)i + 1], zi = x[3 * (size_t)i + 2];
        {
            const double ux = xi - xref[3 * (size_t)i], uy = yi - xref[3 * (size_t)i + 1], uz = zi - xref[3 * (size_t)i + 2];
            const double u2 = ux * ux + uy * uy + uz * uz;
            stale += (u2 <= half_skin2) ? 0.0 : 1.0;  // NaN counts as moved
        }
        double fx = 0.0, fy = 0.0, fz = 0.0;
        constexpr uint32_t G = LJ_EVAL_GROUP;
        const uint32_t n4 = (uint32_t)cnt[i];  // (a multiple of G)
        int32_t jn[G];  // the NEXT group's list entries: loaded one group ahead of the gathers they feed
#pragma unroll
        for (uint32_t u = 0; u < G; ++u) jn[u] = -1;
        // ROWS: the table as the list kernel wrote it, nbr[i * max_nbr + k] -- one 16-byte load brings four entries, but every
        // lane reads a line of its own (+12 % on this kernel); else the transposed table, nbr[k * natoms + i], read coalesced
        const int4* __restrict__ mine = reinterpret_cast<const int4*>(nbr + (ROWS ? (size_t)i * max_nbr : 0));
        auto fetch = [&](const uint32_t k0) {
            if constexpr (ROWS) {
#pragma unroll
                for (uint32_t q = 0; q < G / 4; ++q) {
                    const int4 v = mine[(k0 >> 2) + q];
                    jn[4 * q] = v.x; jn[4 * q + 1] = v.y; jn[4 * q + 2] = v.z; jn[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (uint32_t u = 0; u < G; ++u) jn[u] = nbr[(size_t)(k0 + u) * natoms + i];
            }
        };
        if (n4 > 0) fetch(0);
        for (uint32_t k0 = 0; k0 < n4; k0 += G) {
            int32_t jc[G];
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) jc[u] = jn[u];
            if (k0 + G < n4) fetch(k0 + G);
            double px[G], py[G], pz[G];
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) {  // G gathers in flight
                const size_t jj = (jc[u] < 0) ? (size_t)i : (size_t)jc[u];
                px[u] = x[3 * jj]; py[u] = x[3 * jj + 1]; pz[u] = x[3 * jj + 2];
            }
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) {  // an empty slot or a pair beyond rc adds selected zeros
                const double dx = xi - px[u], dy = yi - py[u], dz = zi - pz[u];
                const double r2 = dx * dx + dy * dy + dz * dz;
                const bool on = (jc[u] >= 0) && (r2 < rc2);
                const double inv2 = lj_rcp(on ? r2 : 1.0);
                const double s6 = inv2 * inv2 * inv2;
                e += on ? 4.0 * (s6 * s6 - s6) : 0.0;                     // pair_energy   lj.rs:22-25
                const double c = 24.0 * (s6 - 2.0 * (s6 * s6)) * inv2;   // pair_gradient/r  lj.rs:28-32, :57-58
                fx += on ? c * dx : 0.0; fy += on ? c * dy : 0.0; fz += on ? c * dz : 0.0;
                e -= on ? eshift : 0.0;
            }
        }
        g[3 * (size_t)i] = fx; g[3 * (size_t)i + 1] = fy; g[3 * (size_t)i + 2] = fz;
        if constexpr (DOT) {  // math.rs:41: acc += x*y, a multiply and an add with roundings of their own
            gd = __dadd_rn(gd, __dmul_rn(fx, dir[3 * (size_t)i]));
            gd = __dadd_rn(gd, __dmul_rn(fy, dir[3 * (size_t)i + 1]));
            gd = __dadd_rn(gd, __dmul_rn(fz, dir[3 * (size_t)i + 2]));
        }
    }
    if constexpr (DOT) {
        double acc[3] = {0.5 * e, stale, gd};
        grid_reduce<3>(acc, red, c0);
    } else {
        double acc[2] = {0.5 * e, stale};
        grid_reduce<2>(acc, red, c0);
    }
}

#pragma clang fp contract(off)

}  // namespace lh
