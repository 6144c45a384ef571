// rust-lbfgs_amd/csrc/lj.hip -- host side of the Lennard-Jones evaluators (lj.h: the kernels): the all-pairs launch geometry,
// the caller-built neighbour table, and LJ_CELLS -- the cell list / Verlet list that the library builds on the device and
// rebuilds whenever an atom has left its skin (examples/lj.rs:20-64 at scale).
#include "ctx.h"
#include "lj.h"


namespace {
// ---- LJ_CELLS: (re)build the Verlet list of ctx->lj_cells from the positions x (lj.h) --------------------------------
int lj_cells_prepare(lbfgs_hip_ctx* ctx, const lbfgs_hip_objective* obj, uint32_t natoms) {
    if (!(obj->cutoff > 0.0) || !(obj->skin > 0.0))
        return fail(ctx, LBFGS_HIP_ERR_ARG, "LJ_CELLS needs cutoff > 0 and skin > 0");
    uint32_t max_nbr = obj->max_nbr ? obj->max_nbr : 128u;
    max_nbr = (max_nbr + LJ_EVAL_GROUP - 1u) / LJ_EVAL_GROUP * LJ_EVAL_GROUP;  // lists are padded to whole groups of the evaluation
    if (!ctx->lj_cells) ctx->lj_cells = new (std::nothrow) LjCells();
    LjCells* lc = ctx->lj_cells;
    if (!lc) return fail(ctx, LBFGS_HIP_ERR_NOMEM, "out of host memory");
    if (lc->natoms == natoms && lc->max_nbr == max_nbr && lc->cutoff == obj->cutoff && lc->skin == obj->skin) return LBFGS_HIP_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t keep_rebuilds = lc->rebuilds, keep_evals = lc->evals;
    lc->release();
    lc->rebuilds = keep_rebuilds; lc->evals = keep_evals;
    lc->natoms = natoms; lc->max_nbr = max_nbr; lc->cutoff = obj->cutoff; lc->skin = obj->skin;
    lc->ncap = std::min<size_t>(2 * (size_t)natoms + 64, (size_t)1024 * LJ_SCAN_TILE);  // (the scan handles 1024 tiles)
#define LJ_ALLOC(ptr, bytes)                                                                                     \
    do {                                                                                                         \
        hipError_t e_ = hipMalloc(&(ptr), (bytes));                                                              \
        if (e_ != hipSuccess) {                                                                                  \
            lc->release();                                                                                       \
            return fail(ctx, LBFGS_HIP_ERR_NOMEM, "LJ_CELLS buffers (%zu bytes): %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                                                                        \
    } while (0)
    const size_t na = natoms ? natoms : 1;
    LJ_ALLOC(lc->nbr, (size_t)max_nbr * na * sizeof(int32_t));
    LJ_ALLOC(lc->nbr_rows, (size_t)max_nbr * na * sizeof(int32_t));
    LJ_ALLOC(lc->cnt, na * sizeof(int32_t));
    LJ_ALLOC(lc->cell_of, na * sizeof(int32_t));
    LJ_ALLOC(lc->slot_of, na * sizeof(int32_t));
    LJ_ALLOC(lc->sorted, na * sizeof(int32_t));
    LJ_ALLOC(lc->counts, (lc->ncap + 1) * sizeof(int32_t));
    LJ_ALLOC(lc->starts, (lc->ncap + 1) * sizeof(int32_t));
    LJ_ALLOC(lc->cursor, (lc->ncap + 1) * sizeof(int32_t));
    LJ_ALLOC(lc->xref, 3 * na * sizeof(double));
    LJ_ALLOC(lc->xs, 4 * na * sizeof(double));  // 32-byte records: x, y, z, index
    LJ_ALLOC(lc->tile_sums, 1024 * sizeof(int32_t));
    LJ_ALLOC(lc->bbox_part, (size_t)LJ_BBOX_GRID * 6 * sizeof(double));
    LJ_ALLOC(lc->overflow, 64);
#undef LJ_ALLOC
    void* hm = nullptr;
    if (hipHostMalloc(&hm, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        lc->release();
        return fail(ctx, LBFGS_HIP_ERR_NOMEM, "LJ_CELLS host word");
    }
    memset(hm, 0, 64);
    void* dv = nullptr;
    if (hipHostGetDevicePointer(&dv, hm, 0) != hipSuccess) {
        (void)hipHostFree(hm);
        lc->release();
        return fail(ctx, LBFGS_HIP_ERR_HIP, "LJ_CELLS host word: no device pointer");
    }
    lc->host = (double*)hm;
    lc->host_dev = (double*)dv;
    return LBFGS_HIP_OK;
}

constexpr uint64_t LJ_TRANSPOSE_AFTER = 6;  // evaluations a list has to serve before its table is turned (0.17 ms / 0.05 ms, rounded up)
void lj_cells_transpose(lbfgs_hip_ctx* ctx, LjCells* lc) {
    hipLaunchKernelGGL(lj_cells_transpose_kernel, dim3(std::min<uint32_t>(4096u, (lc->natoms + 63) / 64)), dim3(BLOCK), 0, ctx->stream,
                       lc->nbr_rows, lc->cnt, lc->natoms, lc->max_nbr, lc->nbr);
    lc->transposed = true;
}

int lj_cells_rebuild(lbfgs_hip_ctx* ctx, const double* x) {
    LjCells* lc = ctx->lj_cells;
    const uint32_t natoms = lc->natoms;
    const double rl = lc->cutoff + lc->skin;
    // 1. bounding box (per-workgroup partials, reduced here: a rebuild synchronises anyway)
    const int bgrid = (int)std::min<uint32_t>(LJ_BBOX_GRID, std::max(1u, (natoms + BLOCK - 1) / BLOCK));
    hipLaunchKernelGGL(lj_bbox_kernel, dim3(bgrid), dim3(BLOCK), 0, ctx->stream, x, natoms, lc->bbox_part);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<double> part((size_t)bgrid * 6);
    HIP_TRY(ctx, hipMemcpyAsync(part.data(), lc->bbox_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = 0; b < bgrid; ++b)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], part[(size_t)b * 6 + k]);
            hi[k] = std::fmax(hi[k], part[(size_t)b * 6 + 3 + k]);
        }
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k]))
            return fail(ctx, LBFGS_HIP_ERR_ARG, "LJ_CELLS: a coordinate is not finite");
    // 2. cells of side >= rl, as many as fit the arrays
    LjGrid gr{};
    double side = rl;
    for (;;) {
        double dims[3], total = 1.0;
        for (int k = 0; k < 3; ++k) {
            dims[k] = std::floor((hi[k] - lo[k]) / side) + 1.0;
            total *= dims[k];
        }
        if (total <= (double)lc->ncap && dims[0] < 2.0e9 && dims[1] < 2.0e9 && dims[2] < 2.0e9) {
            gr.nx = (int)dims[0]; gr.ny = (int)dims[1]; gr.nz = (int)dims[2];
            break;
        }
        side *= 1.26;  // 2x the cell volume
    }
    gr.ox = lo[0]; gr.oy = lo[1]; gr.oz = lo[2];
    gr.inv_side = 1.0 / side;
    const uint32_t ncells = (uint32_t)gr.nx * (uint32_t)gr.ny * (uint32_t)gr.nz;
    // 3. count -> scan -> fill -> sort -> list
    const int agrid = (int)std::min<uint32_t>(4096u, std::max(1u, (natoms + BLOCK - 1) / BLOCK));
    const int cgrid = (int)std::min<uint32_t>(4096u, std::max(1u, (ncells + BLOCK - 1) / BLOCK));
    HIP_TRY(ctx, hipMemsetAsync(lc->counts, 0, ((size_t)ncells + 1) * sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(lc->overflow, 0, sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(lj_cells_count_kernel, dim3(agrid), dim3(BLOCK), 0, ctx->stream, x, natoms, gr, lc->cell_of, lc->slot_of,
                       lc->counts);
    const uint32_t ntiles = (ncells + LJ_SCAN_TILE - 1) / LJ_SCAN_TILE;  // <= 1024 by the choice of ncap
    hipLaunchKernelGGL(lj_scan_tile_sums_kernel, dim3(ntiles), dim3(BLOCK), 0, ctx->stream, lc->counts, ncells, lc->tile_sums);
    hipLaunchKernelGGL(lj_scan_tile_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, lc->tile_sums, ntiles);
    hipLaunchKernelGGL(lj_scan_apply_kernel, dim3(ntiles), dim3(BLOCK), 0, ctx->stream, lc->counts, ncells, lc->tile_sums,
                       lc->starts, lc->cursor);
    hipLaunchKernelGGL(lj_cells_fill_kernel, dim3(agrid), dim3(BLOCK), 0, ctx->stream, lc->cell_of, lc->slot_of, natoms, lc->starts,
                       lc->sorted);
    hipLaunchKernelGGL(lj_cells_sort_kernel, dim3(cgrid), dim3(BLOCK), 0, ctx->stream, lc->starts, ncells, lc->sorted);
    // The candidate tests run in single precision on origin-relative positions against rl^2 + a margin that covers their
    // rounding: |x - o| <= extent, so each fp32 coordinate is off by <= extent * 2^-24, a difference by twice that, and
    // r^2 by <= 2 * sqrt(3) * rl * extent * 2^-23 (+ the fp32 arithmetic's own 1e-6 relative) -- a slightly larger list, the
    // same results (lj.h).  Boxes so large that the margin would exceed 1 % of rl^2 take the double-precision kernel.
    const double extent = std::fmax(std::fmax(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]) + rl;
    const double margin = 8.0 * rl * extent * std::ldexp(1.0, -23) + 2e-6 * rl * rl;
    if (ctx->lj_build_fp32 && margin <= 0.01 * rl * rl) {
        HIP_TRY(ctx, hipMemcpyAsync(lc->xref, x, 3 * (size_t)natoms * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(lj_cells_gather32_kernel, dim3(agrid), dim3(BLOCK), 0, ctx->stream, x, natoms, lc->sorted, gr,
                           reinterpret_cast<LjRec32*>(lc->xs));
        hipLaunchKernelGGL(lj_cells_build32_kernel, dim3(agrid), dim3(BLOCK), 0, ctx->stream,
                           reinterpret_cast<const LjRec32*>(lc->xs), natoms, gr, lc->cell_of, lc->starts, (float)(rl * rl + margin),
                           lc->max_nbr, lc->nbr_rows, lc->cnt, lc->overflow);
    } else {
        hipLaunchKernelGGL(lj_cells_gather_kernel, dim3(agrid), dim3(BLOCK), 0, ctx->stream, x, natoms, lc->sorted, lc->xs);
        hipLaunchKernelGGL(lj_cells_build_kernel, dim3(agrid), dim3(BLOCK), 0, ctx->stream, lc->xs, natoms, gr, lc->cell_of,
                           lc->starts, lc->sorted, rl * rl, lc->max_nbr, lc->nbr_rows, lc->cnt, lc->xref, lc->overflow);
    }
    lc->last_lifetime = lc->evals_this_list;
    lc->evals_this_list = 0;
    lc->transposed = false;
    if (lc->last_lifetime >= LJ_TRANSPOSE_AFTER) lj_cells_transpose(ctx, lc);  // (its predecessor was long-lived)
    HIP_TRY(ctx, hipGetLastError());
    unsigned int over = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&over, lc->overflow, sizeof(over), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (over)
        return fail(ctx, LBFGS_HIP_ERR_ARG, "LJ_CELLS: an atom has %u neighbours within cutoff + skin = %g, the list holds %u "
                    "(raise lbfgs_hip_objective.max_nbr)", over, rl, lc->max_nbr);
    lc->built = true;
    lc->rebuilds += 1;
    return LBFGS_HIP_OK;
}

// `step` != nullptr: a line-search trial -- x is FORMED here as xp + t*d (step->xp, step->d, step->t) by the kernel that checks
// the list, and g.d is summed by the evaluation kernel into *step->dg_out (lbfgs_hip_objective_line_eval)
int lj_cells_eval(lbfgs_hip_ctx* ctx, const lbfgs_hip_objective* obj, const lbfgs_hip_vec* x, lbfgs_hip_vec* g, double* out,
                  uint32_t natoms, const LjTrial* step = nullptr) {
    int rc = lj_cells_prepare(ctx, obj, natoms);
    if (rc != LBFGS_HIP_OK) return rc;
    LjCells* lc = ctx->lj_cells;
    const double half_skin2_early = 0.25 * obj->skin * obj->skin;
    bool formed = false;  // a trial's x = xp + t*d has been written
    if (!lc->built) {
        if (step) {  // (a trial before any evaluation -- a stand-alone Problem: the list is built AT the trial point, so form it first)
            const uint32_t g0 = std::max(1u, std::min((natoms + BLOCK - 1) / BLOCK, 1024u));
            hipLaunchKernelGGL(lj_cells_step_check_kernel, dim3(g0), dim3(BLOCK), 0, ctx->stream, step->xp, step->d, step->t, x->p,
                               step->xp, natoms, half_skin2_early, lc->overflow + 1);
            formed = true;
        }
        if ((rc = lj_cells_rebuild(ctx, x->p)) != LBFGS_HIP_OK) return rc;
    }
    const double rc2 = obj->cutoff * obj->cutoff;
    const double i6 = 1.0 / (rc2 * rc2 * rc2);
    const double eshift = 4.0 * (i6 * i6 - i6);
    const double half_skin2 = 0.25 * obj->skin * obj->skin;
    const uint32_t want = (natoms + BLOCK - 1) / BLOCK;
    const uint32_t grid = std::max(1u, std::min(want, (uint32_t)MAX_GRID));
    for (int attempt = 0;; ++attempt) {
        RedCtl red{};
        if ((rc = fill_handoff(ctx, red, step ? 3 : 2)) != LBFGS_HIP_OK) return rc;
        red.out[0] = out;
        red.out[1] = lc->host_dev;  // the "moved too far" count goes straight to host-mapped memory
        if (step) red.out[2] = step->dg_out;
        // a cheap look first: has the list gone stale?  (then the evaluation kernel returns at once and the list is rebuilt)
        unsigned int* moved_flag = lc->overflow + 1;  // (a word of the 64-byte scratch next to the overflow counter)
        HIP_TRY(ctx, hipMemsetAsync(moved_flag, 0, sizeof(unsigned int), ctx->stream));
        if (step && attempt == 0 && !formed)  // (the trial point is formed by the same pass; after a rebuild x exists already)
            hipLaunchKernelGGL(lj_cells_step_check_kernel, dim3(std::min(grid, 1024u)), dim3(BLOCK), 0, ctx->stream, step->xp,
                               step->d, step->t, x->p, lc->xref, natoms, half_skin2, moved_flag);
        else
            hipLaunchKernelGGL(lj_cells_check_kernel, dim3(std::min(grid, 1024u)), dim3(BLOCK), 0, ctx->stream, x->p, lc->xref, natoms,
                               half_skin2, moved_flag);
        // Which table?  The list kernel writes it row-major; turning it costs 0.17 ms at 1e6 atoms and makes every evaluation
        // 0.05 ms cheaper.  A list that is rebuilt after two or three evaluations (a system that relaxes fast) is read as
        // written; one that has served LJ_TRANSPOSE_AFTER evaluations -- or whose predecessor did -- is turned.
        if (!lc->transposed && lc->evals_this_list >= LJ_TRANSPOSE_AFTER) lj_cells_transpose(ctx, lc);
        const double* dir = step ? step->d : nullptr;
        if (lc->transposed && step)
            hipLaunchKernelGGL((lj_cells_eval_kernel<false, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, x->p, g->p, lc->nbr, lc->cnt,
                               lc->xref, natoms, lc->max_nbr, rc2, eshift, half_skin2, moved_flag, red, dir);
        else if (lc->transposed)
            hipLaunchKernelGGL((lj_cells_eval_kernel<false, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, x->p, g->p, lc->nbr, lc->cnt,
                               lc->xref, natoms, lc->max_nbr, rc2, eshift, half_skin2, moved_flag, red, dir);
        else if (step)
            hipLaunchKernelGGL((lj_cells_eval_kernel<true, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, x->p, g->p, lc->nbr_rows,
                               lc->cnt, lc->xref, natoms, lc->max_nbr, rc2, eshift, half_skin2, moved_flag, red, dir);
        else
            hipLaunchKernelGGL((lj_cells_eval_kernel<true, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, x->p, g->p, lc->nbr_rows,
                               lc->cnt, lc->xref, natoms, lc->max_nbr, rc2, eshift, half_skin2, moved_flag, red, dir);
        HIP_TRY(ctx, hipGetLastError());
        lc->evals += 1;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const double moved = *(volatile double*)lc->host;
        if (moved == 0.0) {
            lc->evals_this_list += 1;
            return LBFGS_HIP_OK;
        }
        if (attempt == 1) return fail(ctx, LBFGS_HIP_ERR_HIP, "LJ_CELLS: the list is stale right after a rebuild");
        // some atom left its skin/2 sphere (or x holds a NaN: the rebuild reports that): rebuild at x, evaluate again
        if ((rc = lj_cells_rebuild(ctx, x->p)) != LBFGS_HIP_OK) return rc;
    }
}

}  // namespace

// Lennard-Jones objectives (lj.h): x holds 3*natoms coordinates of ONE rank
int lj_eval(lbfgs_hip_ctx* ctx, const lbfgs_hip_objective* obj, const lbfgs_hip_vec* x, lbfgs_hip_vec* g, double* out,
            const LjTrial* step) {
    if (ctx->shard.world != 1) return fail(ctx, LBFGS_HIP_ERR_ARG, "the LJ objectives need all atoms on one rank");
    const uint64_t n = ctx->shard.n_local;
    if (n % 3 != 0 || n / 3 > 0x7fffffffULL) return fail(ctx, LBFGS_HIP_ERR_ARG, "LJ needs n = 3*natoms");
    const uint32_t natoms = (uint32_t)(n / 3);
    for (const double* o : {(const double*)out, step ? (const double*)step->dg_out : (const double*)nullptr}) {
        const long idx = o ? o - ctx->board : -1;
        if (idx >= 0 && idx < LBFGS_HIP_BOARD_SLOTS + 2) ctx->mirror_valid[idx] = false;
    }
    if (obj->kind == LBFGS_HIP_OBJ_LJ_CELLS) {
        ProfScope ps(ctx, LBFGS_HIP_K_EVAL);
        return lj_cells_eval(ctx, obj, x, g, out, natoms, step);
    }
    if (step) return fail(ctx, LBFGS_HIP_ERR_ARG, "only LJ_CELLS has a fused trial");
    RedCtl red{};
    {
        const int rc_h = fill_handoff(ctx, red, 1);
        if (rc_h != LBFGS_HIP_OK) return rc_h;
    }
    red.out[0] = out;
    {
        ProfScope ps(ctx, LBFGS_HIP_K_EVAL);
        if (obj->kind == LBFGS_HIP_OBJ_LJ_ALLPAIRS) {
            const uint32_t tiles = std::max(1u, (natoms + BLOCK - 1) / BLOCK);
            if (tiles > (uint32_t)MAX_GRID) return fail(ctx, LBFGS_HIP_ERR_ARG, "all-pairs LJ supports up to %d atoms", MAX_GRID * BLOCK);
            // about six workgroups per CU: enough waves per SIMD to hide the division chains, and a short tail
            uint32_t S = (uint32_t)std::lround(6.0 * ctx->cu_count / tiles);
            S = std::min({std::max(S, 1u), 16u, tiles, (uint32_t)MAX_GRID / tiles});
            const uint32_t jspan = ((natoms + S - 1) / S + BLOCK - 1) / BLOCK * BLOCK;
            S = (natoms + jspan - 1) / jspan;  // ranges that actually hold atoms
            double* out_f = g->p;
            if (S > 1) {
                const size_t need = (size_t)S * n * sizeof(double);
                if (need > ctx->lj_scratch_bytes) {
                    if (ctx->lj_scratch) HIP_TRY(ctx, hipFree(ctx->lj_scratch));
                    ctx->lj_scratch = nullptr;
                    ctx->lj_scratch_bytes = 0;
                    hipError_t e = hipMalloc(&ctx->lj_scratch, need);
                    if (e != hipSuccess) return fail(ctx, LBFGS_HIP_ERR_NOMEM, "LJ force slices: %s", hipGetErrorString(e));
                    ctx->lj_scratch_bytes = need;
                }
                out_f = ctx->lj_scratch;
            }
            hipLaunchKernelGGL(lj_allpairs_kernel, dim3(tiles * S), dim3(BLOCK), 0, ctx->stream, x->p, out_f, natoms, tiles,
                               jspan, red);
            if (S > 1) {
                const uint32_t cg = (uint32_t)std::min<uint64_t>((n + BLOCK - 1) / BLOCK, 1024);
                hipLaunchKernelGGL(lj_combine_kernel, dim3(cg), dim3(BLOCK), 0, ctx->stream, ctx->lj_scratch, g->p, n, S);
            }
        } else {
            if (!obj->nbr_index || obj->max_nbr == 0 || !(obj->cutoff > 0.0))
                return fail(ctx, LBFGS_HIP_ERR_ARG, "LJ_NEIGHBORS needs a neighbour table and a cutoff");
            const double rc2 = obj->cutoff * obj->cutoff;
            const double i6 = 1.0 / (rc2 * rc2 * rc2);
            const double eshift = 4.0 * (i6 * i6 - i6);
            const uint32_t want = (natoms + BLOCK - 1) / BLOCK;
            const uint32_t grid = want < (uint32_t)MAX_GRID ? want : (uint32_t)MAX_GRID;
            hipLaunchKernelGGL(lj_neighbors_kernel, dim3(grid ? grid : 1), dim3(BLOCK), 0, ctx->stream, x->p, g->p,
                               (const int32_t*)obj->nbr_index, obj->max_nbr, natoms, rc2, eshift, red);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return LBFGS_HIP_OK;
}


extern "C" {

int lbfgs_hip_lj_cells_stats(lbfgs_hip_ctx* ctx, uint64_t* rebuilds, uint64_t* evaluations, uint32_t* longest_list) {
    if (!ctx) return LBFGS_HIP_ERR_ARG;
    const LjCells* lc = ctx->lj_cells;
    uint32_t longest = 0;
    if (lc && lc->built && lc->natoms) {  // longest list of the latest build (a small scan, diagnostics only)
        std::vector<int32_t> cnt(lc->natoms);
        HIP_TRY(ctx, hipMemcpyAsync(cnt.data(), lc->cnt, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int32_t c : cnt) longest = std::max(longest, (uint32_t)c);
    }
    if (rebuilds) *rebuilds = lc ? lc->rebuilds : 0;
    if (evaluations) *evaluations = lc ? lc->evals : 0;
    if (longest_list) *longest_list = longest;
    return LBFGS_HIP_OK;
}

}  // extern "C"
