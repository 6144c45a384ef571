// rust-lbfgs_amd/csrc/lbfgs_hip.hip -- the launchers behind include/lbfgs_hip.h for gfx950 (MI355X): one wrapper per C-ABI
// entry point of the hot path -- BLAS-1 primitives, fused operators, history update, the two-loop recursion in all its
// forms (persistent on-chip kernel, kernel per step, vector-free), OWL-QN operators, element-wise
// objectives.  The kernels are instantiations of the streaming skeleton in stream.h over the element operators in ops.h,
// plus the persistent kernels of resident.h and gram_combine.h.  Context, communicators, vectors and scalar reads live in
// context.hip, the Lennard-Jones bookkeeping in lj.hip (ctx.h is what the three share).
#include "ctx.h"

#include "ops.h"
#include "gram.h"
#include "resident.h"
#include "gram_combine.h"

namespace {
#ifndef LH_NT_IN   // cache-hint masks of the streaming instantiation (bit s = stream s uses `nt`); A/B builds override them
#define LH_NT_IN ~0u
#endif
#ifndef LH_NT_OUT
#define LH_NT_OUT ~0u
#endif
// ---- launch one operator ---------------------------------------------------------------------
// partials_only (Op::NRED == 1, single rank): the kernel leaves its workgroups' partial sums in red_out[0][0 .. grid) and
// nobody reduces them -- the consumer does (stream.h RedCtl::tagged == 2); *grid_out receives the number of partials.
// ALIASING CONTRACT: an output stream may BE an input stream (y += c x; OpObjOwlLineEval<Obj, true> reads the previous
// pseudo-gradient through in[2] and writes the new one through out[2], the same buffer).  That is sound because the skeleton
// (stream.h stream_sweep) loads ALL inputs of an element -- of a whole trip -- before it stores any output of that element, every
// element is touched by exactly one thread, and neither in[] nor out[] is declared __restrict__; the `nt` hint is a cache
// policy, not a promise of no aliasing.  An operator must not read, for element i, an input that aliases an output at another
// index j != i.  Held by test_fused_owlqn_kernels_equal_their_unfused_sequences (bitwise, odd n, every cache-hint regime).
template <class Op>
int launch(lbfgs_hip_ctx* ctx, int kclass, const Op& op, double* const* red_out, double* dup_ptr = nullptr,
           int dup_k = 0, bool partials_only = false, unsigned int* grid_out = nullptr) {
    static_assert(Op::NRED <= MAX_RED, "the partials buffer holds MAX_RED sums per workgroup");
    ctx->last_res.valid = false;
    RedCtl red{};
    bool in_kernel_exchange = false;
    if (partials_only) {
        red.tagged = 2u;
        red.ctr = ctx->dev_ctr;
        for (int k = 0; k < Op::NRED && k < RED_PTRS; ++k) red.out[k] = red_out[k];
    } else {
        const int rc_p = prep_red(ctx, red, Op::NRED, red_out, dup_ptr, dup_k, &in_kernel_exchange);
        if (rc_p != LBFGS_HIP_OK) return rc_p;
    }
    const uint64_t n = ctx->shard.n_local;
    constexpr int MAP = tuning<Op>::MAP, UNR = tuning<Op>::UNR;
    const int x32_k = (kclass >= 0 && kclass < LBFGS_HIP_K_CLASSES) ? ctx->grid_x32_class[kclass] : 0;
    const int grid = grid_for(ctx, x32_k > 0 ? x32_k : tuning<Op>::GRID_X32);
    if (grid_out) *grid_out = (unsigned int)grid;
    // Cache hints by vector size (profiles/r02_shard_cache_hints.log):
    //   >= 128 MiB  `nt` (streaming) on loads and stores: nothing can stay in the 256 MiB Infinity Cache anyway;
    //   64-128 MiB  `nt` on the stores only: the loads keep finding the running vector in the Infinity Cache, while
    //               streamed stores leave fewer dirty lines in the L2s to write back when the kernel ends (+2.3 % on
    //               whole iterations at 100 MB shards, +1.2 % at 80 MB, nothing at 24 MB, -2.5 % at 8 MB);
    //   below       plain.
    const size_t vbytes = n * sizeof(double);
    const bool streaming = vbytes >= ctx->nt_threshold_bytes;
    const bool stream_stores = !streaming && vbytes >= ctx->nt_store_threshold_bytes && Op::NOUT > 0;
    {
        ProfScope ps(ctx, kclass);
        if (streaming)
            hipLaunchKernelGGL((stream_kernel<Op, UNR, LH_NT_IN, LH_NT_OUT, MAP>), dim3(grid), dim3(BLOCK), 0, ctx->stream,
                               op, n, ctx->shard.offset, red);
        else if (stream_stores)
            hipLaunchKernelGGL((stream_kernel<Op, UNR, 0u, LH_NT_OUT, MAP, 1, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream,
                               op, n, ctx->shard.offset, red);
        else  // (the two smaller regimes issue the first trip's loads before Op::setup(): stream.h LATE)
            hipLaunchKernelGGL((stream_kernel<Op, UNR, 0u, 0u, MAP, 1, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream,
                               op, n, ctx->shard.offset, red);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (Op::NRED > 0 && !in_kernel_exchange && !partials_only) return allreduce(ctx, red_out, Op::NRED);
    return LBFGS_HIP_OK;
}
}  // namespace

// ---- vector-free (Gram) two-loop: scalar recursion on the coefficients --------------------------------
constexpr int GRAM_MAX_M = 10;  // 3*(2m+1) sums must fit MAX_RED
constexpr int GRAM_MAX_NB = 2 * GRAM_MAX_M + 1;
struct GramArgs {
    double* G;             // nb x nb, basis order [s_0..s_{m-1}, y_0..y_{m-1}, g]
    const double* rows;    // 3 x nb in POSITION order of the rows kernel
    double* delta;         // nb coefficients out (basis order)
    double* pred;          // out [2]: [0] ||d||^2 = delta^T G delta as the GRAM arithmetic sees it (NaN if a coefficient is not
                           // finite); the combine pass sums the real ||d||^2 next to it.  [1] the CANCELLATION of the recursion's
                           // running vector in the basis: max over its two turning points (after the first loop, at the end) of
                           // sum_t ||b_t|| |delta_t| / ||sum_t delta_t b_t||.  A dot product s_j.q taken through Gram entries
                           // is as accurate as its largest terms, the exact recursion's as the vector q itself: the ratio is
                           // how many times more rounding error the coefficient-space recursion carries than the exact one
                           // (lbfgs_hip_two_loop_gram, board[dnorm_slot + 2 .. + 3])
    double* alpha;         // history alpha[m] (kept for API parity, lbfgs.rs:587)
    const double* ys;      // history ys[m]: the STORED y.s (lbfgs.rs:656) -- under Powell damping it is the
                           // pre-damping value and differs from the Gram entry of the damped y
    const double *gnum, *gden;
    int m, bound;
    int row_basis[3];                 // basis index of the three refreshed rows
    int pos_to_basis[GRAM_MAX_NB];    // position p of the rows kernel -> basis index
    int order[GRAM_MAX_M];            // slots, newest first (lbfgs.rs:583)
};

__global__ __launch_bounds__(64) void gram_coef_kernel(const GramArgs a) {
    __shared__ double G[GRAM_MAX_NB * GRAM_MAX_NB];
    __shared__ double delta[GRAM_MAX_NB];
    __shared__ double alpha[GRAM_MAX_M];
    const int nb = 2 * a.m + 1, lane = threadIdx.x;
    // 1. refresh the three rows/columns that changed, then stage G in LDS
    for (int i = lane; i < 3 * nb; i += 64) {
        const int r = i / nb, p = i % nb;
        const int bi = a.row_basis[r], bj = a.pos_to_basis[p];
        const double v = a.rows[i];
        a.G[bi * nb + bj] = v;
        a.G[bj * nb + bi] = v;
    }
    __syncthreads();
    for (int i = lane; i < nb * nb; i += 64) G[i] = a.G[i];
    __syncthreads();
    if (lane == 0) {
        const int m = a.m;
        for (int j = 0; j < nb; ++j) delta[j] = 0.0;
        delta[2 * m] = -1.0;  // q = -g
        for (int it = 0; it < a.bound; ++it) {  // lbfgs.rs:582-590
            const int j = a.order[it];
            double dot = 0.0;
            for (int t = 0; t < nb; ++t) dot += G[j * nb + t] * delta[t];        // s_j . q
            const double al = dot / a.ys[j];                                      // / ys_j (stored, lbfgs.rs:587)
            alpha[j] = al;
            delta[m + j] = delta[m + j] + (-al);                                  // q -= alpha*y_j
        }
        auto cancellation = [&]() {  // (see GramArgs::pred)
            double s1 = 0.0, qq = 0.0;
            for (int i = 0; i < nb; ++i) {
                double row = 0.0;
                for (int t = 0; t < nb; ++t) row += G[i * nb + t] * delta[t];
                qq += delta[i] * row;
                s1 += sqrt(fmax(G[i * nb + i], 0.0)) * fabs(delta[i]);
            }
            return s1 / sqrt(qq);  // (qq <= 0 through rounding: inf or NaN, i.e. "no digits left")
        };
        const double cancel_mid = cancellation();
        const double gamma = *a.gnum / *a.gden;                                   // lbfgs.rs:591
        for (int t = 0; t < nb; ++t) delta[t] = delta[t] * gamma;
        for (int it = a.bound - 1; it >= 0; --it) {  // lbfgs.rs:594-601
            const int j = a.order[it];
            double dot = 0.0;
            for (int t = 0; t < nb; ++t) dot += G[(m + j) * nb + t] * delta[t];  // y_j . q
            const double beta = dot / a.ys[j];                                    // lbfgs.rs:597
            delta[j] = delta[j] + (alpha[j] - beta);                              // q += (alpha-beta)*s_j
        }
        for (int t = 0; t < nb; ++t) a.delta[t] = delta[t];
        for (int it = 0; it < a.bound; ++it) a.alpha[a.order[it]] = alpha[a.order[it]];
        double dd = 0.0;  // delta^T G delta, row by row
        bool finite = true;
        for (int i = 0; i < nb; ++i) {
            double row = 0.0;
            for (int t = 0; t < nb; ++t) row += G[i * nb + t] * delta[t];
            dd += delta[i] * row;
            finite = finite && (delta[i] - delta[i] == 0.0);
        }
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        const double cancel_end = cancellation();
        a.pred[0] = finite ? dd : nan;
        a.pred[1] = (cancel_mid == cancel_mid && cancel_end == cancel_end) ? fmax(cancel_mid, cancel_end) : nan;
    }
}

// ---- helpers of the C-ABI functions below (templates need C++ linkage) -----------------------
namespace {
// A dot product on its way from the kernel that sums it to the kernel that consumes it: either a finished scalar
// (board slot), or -- single rank, tagged hand-off -- the producer's workgroup partials, left for the consumer to add up.
struct DotRef {
    double* scalar = nullptr;
    double* parts = nullptr;
    unsigned int nparts = 0;  // > 0: `parts` holds that many partials; else `scalar` holds the sum
};

template <bool NEG_SRC, bool SCALE, int VMODE>
int two_loop_step(lbfgs_hip_history* h, const double* src, const double* u, const double* v, double* dst,
                  const DotRef& dot_in, int j, int mode_b, const double* gnum, const double* gden, DotRef* dot_out,
                  int kclass, uint64_t owl_start = 0, uint64_t owl_end = 0) {
    OpTwoLoopStep<NEG_SRC, SCALE, VMODE> op{};
    op.in[0] = src; op.in[1] = u; op.in[2] = v;
    op.out[0] = dst;
    op.dot_in = dot_in.scalar;
    op.dot_parts = dot_in.parts;
    op.dot_nparts = dot_in.nparts;
    op.ys_j = h->ys + j;
    op.alpha_j = h->alpha + j;
    op.gamma_num = gnum;
    op.gamma_den = gden;
    op.mode_b = mode_b;
    op.owl_start = owl_start;
    op.owl_end = owl_end;
    if (dot_out->parts) {  // (single-sum kernels only: VMODE 0 / 1)
        double* outs[1] = {dot_out->parts};
        return launch(h->ctx, kclass, op, outs, nullptr, 0, true, &dot_out->nparts);
    }
    double* out = dot_out->scalar;
    double* outs[4] = {out, out + 1, out + 2, out + 3};  // VMODE 2 / 3 produce 2 / 4 adjacent sums
    return launch(h->ctx, kclass, op, outs);
}
}  // namespace

namespace {
int rosen_shape_ok(lbfgs_hip_ctx* ctx) {
    if ((ctx->shard.n_local & 1) || (ctx->shard.offset & 1))
        return fail(ctx, LBFGS_HIP_ERR_ARG, "Rosenbrock couples (x[2i], x[2i+1]): shard size and offset must be even");
    return LBFGS_HIP_OK;
}
}  // namespace

namespace {
// the rows pass with the row vectors' tiles on the chip (gram_combine.h): -> true if launched (red is set up by the caller)
bool gram_rows_resident(lbfgs_hip_ctx* ctx, const double* const* in, int nb, const RedCtl& red) {
    if (!ctx->resident_on || !ctx->gram_combine_resident || ctx->grid_override > 0 ||
        nb > GC_MAX_COLS || 3 * nb > MAX_RED)
        return false;
    const uint64_t n = ctx->shard.n_local;
    const int grid = std::min(ctx->cu_count, (int)BLOCK);
    const uint64_t per_round = (uint64_t)grid * BLOCK;
    if (n < 2 || (n >> 1) + per_round * (GR_T + 1) >= (1ull << 28)) return false;
    const size_t lds_bytes = (size_t)3 * nb * BLOCK * sizeof(double);
    if (lds_bytes > (size_t)150 * 1024) return false;  // (m <= 11: the sums of a workgroup must fit its LDS next to the static part)
    const bool nt = n * sizeof(double) >= ctx->resident_nt_bytes;
    auto kern = nt ? gram_rows_resident_kernel<true> : gram_rows_resident_kernel<false>;
    const unsigned long long bit = 1ull << (42 + (nt ? 1 : 0));
    if (!(ctx->resident_attr_mask & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=
            hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        ctx->resident_attr_mask |= bit;
    }
    GramRowsResArgs a{};
    for (int j = 0; j < nb; ++j) a.in[j] = in[j];
    a.n = n; a.nb = nb;
    a.total_rounds = (uint32_t)(((n >> 1) + per_round - 1) / per_round);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds_bytes, ctx->stream, a, red);
    return true;
}

// -> 1 launched, 0 not eligible (the caller streams all columns at once), < 0 error.  dn: ||d||^2, g.d
int gram_combine_resident(lbfgs_hip_ctx* ctx, const double* const* cols, int nb, double* d, const double* delta, const double* pred,
                          double* dn) {
    if (!ctx->resident_on || !ctx->gram_combine_resident || ctx->handoff_ticket || ctx->grid_override > 0 ||
        nb > GC_MAX_COLS)
        return 0;
    const uint64_t n = ctx->shard.n_local;
    const int grid = std::min(ctx->cu_count, (int)BLOCK);
    const uint64_t per_round = (uint64_t)grid * BLOCK;
    if (n < 2 || (n >> 1) + per_round >= (1ull << 28)) return 0;
    const uint64_t total = ((n >> 1) + per_round - 1) / per_round;
    constexpr uint64_t TILE_MAX = GC_ER + RES_LDS_PAIRS_MAX;
    const uint64_t ntiles = (total + TILE_MAX - 1) / TILE_MAX;
    const uint64_t tile = (total + ntiles - 1) / ntiles;  // balanced tiles of <= 96 rounds
    const size_t lds_bytes = (size_t)RES_LDS_PAIRS_MAX * BLOCK * sizeof(d2);
    const bool nt = n * sizeof(double) >= ctx->resident_nt_bytes;
    auto kern = nt ? gram_combine_resident_kernel<true> : gram_combine_resident_kernel<false>;
    const unsigned long long bit = 1ull << (40 + (nt ? 1 : 0));
    if (!(ctx->resident_attr_mask & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) !=
            hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        ctx->resident_attr_mask |= bit;
    }
    GramCombArgs a{};
    for (int j = 0; j < nb; ++j) a.in[j] = cols[j];
    a.d = d; a.delta = delta; a.n = n; a.nb = nb;
    a.pred = pred;
    // the rank that owns global element 0 carries the prediction into the third and fourth sums (this kernel only runs on
    // shards of at least two elements; an EMPTY rank 0 takes the streaming combine, which keys on the element index itself)
    a.lead = ctx->shard.offset == 0 ? 1 : 0;
    a.total_rounds = (uint32_t)total;
    a.tile_rounds = (uint32_t)tile;
    RedCtl red{};
    bool in_kernel_exchange = false;
    double* outs2[4] = {dn, dn + 1, dn + 2, dn + 3};
    const int rc_p = prep_red(ctx, red, 4, outs2, nullptr, 0, &in_kernel_exchange);
    if (rc_p != LBFGS_HIP_OK) return rc_p;
    {
        ProfScope ps(ctx, LBFGS_HIP_K_TWOLOOP_STEP);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds_bytes, ctx->stream, a, red);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (!in_kernel_exchange) {
        const int rc = allreduce(ctx, outs2, 4);
        if (rc != LBFGS_HIP_OK) return rc;
    }
    return 1;
}

template <int M>
int two_loop_gram_impl(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, int bound, int end, int e1,
                       const double* gnum, const double* gden, double* dn) {
    lbfgs_hip_ctx* ctx = h->ctx;
    constexpr int NB = 2 * M + 1;
    GramArgs ga{};
    // rows kernel: position 0 = new s, 1 = new y, 2 = g, then every other basis vector in ascending order
    GramRowsArgs<M> rows{};
    rows.in[0] = h->s[end]->p; rows.in[1] = h->y[end]->p; rows.in[2] = g->p;
    ga.pos_to_basis[0] = end; ga.pos_to_basis[1] = M + end; ga.pos_to_basis[2] = 2 * M;
    int p = 3;
    for (int b = 0; b < 2 * M; ++b) {
        if (b == end || b == M + end) continue;
        rows.in[p] = (b < M) ? h->s[b]->p : h->y[b - M]->p;
        ga.pos_to_basis[p] = b;
        ++p;
    }
    {
        RedCtl red{};
        {
            const int rc_h = fill_handoff(ctx, red, 3 * NB);
            if (rc_h != LBFGS_HIP_OK) return rc_h;
        }
        red.out_contig = h->gram_rows;
        if (ctx->comm_kind == LBFGS_HIP_COMM_P2P) red.p2p = next_p2p(ctx);
        const uint64_t n = ctx->shard.n_local;
        const int grid = ctx->gram_grid > 0 ? ctx->gram_grid : grid_for(ctx);
        ProfScope ps(ctx, LBFGS_HIP_K_TWOLOOP_STEP);
        if (gram_rows_resident(ctx, rows.in, NB, red)) {
            // (the three row vectors' tiles on the chip, the other columns streamed past them: gram_combine.h)
        } else if (n * sizeof(double) >= ctx->nt_threshold_bytes)
            hipLaunchKernelGGL((gram_rows_kernel<M, true, GRAM_ROWS_UNROLL>), dim3(grid), dim3(BLOCK), 0, ctx->stream, rows, n, red);
        else
            hipLaunchKernelGGL((gram_rows_kernel<M, false, GRAM_ROWS_UNROLL>), dim3(grid), dim3(BLOCK), 0, ctx->stream, rows, n, red);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->comm_kind != LBFGS_HIP_COMM_P2P) {
        std::vector<double*> ptrs(3 * NB);
        for (int i = 0; i < 3 * NB; ++i) ptrs[i] = h->gram_rows + i;
        int rc = allreduce(ctx, ptrs.data(), 3 * NB);
        if (rc != LBFGS_HIP_OK) return rc;
    }
    // scalar recursion on the coefficients
    ga.G = h->gram; ga.rows = h->gram_rows; ga.delta = h->gram_delta; ga.pred = h->gram_pred; ga.alpha = h->alpha; ga.ys = h->ys;
    ga.gnum = gnum; ga.gden = gden; ga.m = M; ga.bound = bound;
    ga.row_basis[0] = end; ga.row_basis[1] = M + end; ga.row_basis[2] = 2 * M;
    for (int i = 0; i < bound; ++i) ga.order[i] = ((e1 - 1 - i) % M + M) % M;
    {
        ProfScope ps(ctx, LBFGS_HIP_K_TWOLOOP_EDGE);
        hipLaunchKernelGGL(gram_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, ga);
    }
    HIP_TRY(ctx, hipGetLastError());
    // d = sum_j delta_j b_j, ||d||^2, g.d -- with d accumulated on the chip, one column at a time (gram_combine.h), or, where
    // that kernel cannot run, as a streaming pass over all the columns at once
    {
        const double* cols[2 * M + 1];
        for (int b = 0; b < M; ++b) { cols[b] = h->s[b]->p; cols[M + b] = h->y[b]->p; }
        cols[2 * M] = g->p;
        const int rr = gram_combine_resident(ctx, cols, 2 * M + 1, d->p, h->gram_delta, h->gram_pred, dn);
        if (rr != 0) return rr < 0 ? rr : LBFGS_HIP_OK;
    }
    OpGramCombine<M> cmb{};
    for (int b = 0; b < M; ++b) { cmb.in[b] = h->s[b]->p; cmb.in[M + b] = h->y[b]->p; }
    cmb.in[2 * M] = g->p;
    cmb.out[0] = d->p;
    cmb.delta = h->gram_delta;
    cmb.pred = h->gram_pred;
    double* outs2[4] = {dn, dn + 1, dn + 2, dn + 3};
    return launch(ctx, LBFGS_HIP_K_TWOLOOP_STEP, cmb, outs2);
}
}  // namespace

static int two_loop_impl(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                         int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                         bool owl, uint64_t owl_start, uint64_t owl_end);

extern "C" {

// ==================================================================================== primitives
int lbfgs_hip_vec_fill(lbfgs_hip_vec* v, double value) {
    if (!v) return LBFGS_HIP_ERR_ARG;
    OpFill op{};
    op.out[0] = v->p;
    op.c = value;
    return launch(v->ctx, LBFGS_HIP_K_BLAS1, op, nullptr);
}

int lbfgs_hip_vecadd(lbfgs_hip_vec* y, const lbfgs_hip_vec* x, double c) {
    if (!same_ctx(y, x)) return LBFGS_HIP_ERR_ARG;
    OpAxpy op{};
    op.in[0] = y->p; op.in[1] = x->p; op.out[0] = y->p;
    op.c_host = c; op.c_dev = nullptr;
    return launch(y->ctx, LBFGS_HIP_K_BLAS1, op, nullptr);
}

int lbfgs_hip_vecadd_dev(lbfgs_hip_vec* y, const lbfgs_hip_vec* x, int c_slot) {
    if (!same_ctx(y, x) || !slot_ok(c_slot, 1)) return LBFGS_HIP_ERR_ARG;
    OpAxpy op{};
    op.in[0] = y->p; op.in[1] = x->p; op.out[0] = y->p;
    op.c_host = 0.0; op.c_dev = y->ctx->board + c_slot;
    return launch(y->ctx, LBFGS_HIP_K_BLAS1, op, nullptr);
}

int lbfgs_hip_vecdot(const lbfgs_hip_vec* x, const lbfgs_hip_vec* y, int out_slot) {
    if (!same_ctx(x, y) || !slot_ok(out_slot, 1)) return LBFGS_HIP_ERR_ARG;
    OpDot op{};
    op.in[0] = x->p; op.in[1] = y->p;
    double* outs[1] = {x->ctx->board + out_slot};
    return launch(x->ctx, LBFGS_HIP_K_BLAS1, op, outs);
}

int lbfgs_hip_vecscale(lbfgs_hip_vec* y, double c) {
    if (!y) return LBFGS_HIP_ERR_ARG;
    OpScale op{};
    op.in[0] = y->p; op.out[0] = y->p; op.c = c;
    return launch(y->ctx, LBFGS_HIP_K_BLAS1, op, nullptr);
}

int lbfgs_hip_veccpy(lbfgs_hip_vec* y, const lbfgs_hip_vec* x) {
    if (!same_ctx(y, x)) return LBFGS_HIP_ERR_ARG;
    OpCopy<false> op{};
    op.in[0] = x->p; op.out[0] = y->p;
    return launch(y->ctx, LBFGS_HIP_K_BLAS1, op, nullptr);
}

int lbfgs_hip_vecncpy(lbfgs_hip_vec* y, const lbfgs_hip_vec* x) {
    if (!same_ctx(y, x)) return LBFGS_HIP_ERR_ARG;
    OpCopy<true> op{};
    op.in[0] = x->p; op.out[0] = y->p;
    return launch(y->ctx, LBFGS_HIP_K_BLAS1, op, nullptr);
}

int lbfgs_hip_vecdiff(lbfgs_hip_vec* z, const lbfgs_hip_vec* x, const lbfgs_hip_vec* y) {
    if (!same_ctx(z, x) || !same_ctx(z, y)) return LBFGS_HIP_ERR_ARG;
    OpDiff op{};
    op.in[0] = x->p; op.in[1] = y->p; op.out[0] = z->p;
    return launch(z->ctx, LBFGS_HIP_K_BLAS1, op, nullptr);
}

int lbfgs_hip_vec2norm_sq(const lbfgs_hip_vec* x, int out_slot) {
    if (!x || !slot_ok(out_slot, 1)) return LBFGS_HIP_ERR_ARG;
    OpNrm2 op{};
    op.in[0] = x->p;
    double* outs[1] = {x->ctx->board + out_slot};
    return launch(x->ctx, LBFGS_HIP_K_BLAS1, op, outs);
}

int lbfgs_hip_vec2norm(const lbfgs_hip_vec* x, int scratch_slot, double* out) {  // math.rs:73-76
    if (!x || !out || !slot_ok(scratch_slot, 1)) return LBFGS_HIP_ERR_ARG;
    int rc = lbfgs_hip_vec2norm_sq(x, scratch_slot);
    if (rc != LBFGS_HIP_OK) return rc;
    double sq = 0.0;
    if ((rc = lbfgs_hip_scalars_read(x->ctx, scratch_slot, 1, &sq)) != LBFGS_HIP_OK) return rc;
    *out = std::sqrt(sq);
    return LBFGS_HIP_OK;
}

int lbfgs_hip_vec2norminv(const lbfgs_hip_vec* x, int scratch_slot, double* out) {  // math.rs:79-81
    double nrm = 0.0;
    const int rc = lbfgs_hip_vec2norm(x, scratch_slot, &nrm);
    if (rc != LBFGS_HIP_OK) return rc;
    *out = 1.0 / nrm;
    return LBFGS_HIP_OK;
}

// ==================================================================================== fused ops
int lbfgs_hip_line_step(lbfgs_hip_vec* x, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double step,
                        const lbfgs_hip_vec* wp, uint64_t start, uint64_t end) {
    if (!same_ctx(x, xp) || !same_ctx(x, d) || (wp && !same_ctx(x, wp))) return LBFGS_HIP_ERR_ARG;
    if (wp) {
        OpLineStep<true> op{};
        op.in[0] = xp->p; op.in[1] = d->p; op.in[2] = wp->p; op.out[0] = x->p;
        op.step = step; op.start = start; op.end = end;
        return launch(x->ctx, LBFGS_HIP_K_LINE, op, nullptr);
    }
    OpLineStep<false> op{};
    op.in[0] = xp->p; op.in[1] = d->p; op.in[2] = nullptr; op.out[0] = x->p;
    op.step = step; op.start = 0; op.end = 0;
    return launch(x->ctx, LBFGS_HIP_K_LINE, op, nullptr);
}

int lbfgs_hip_norms_sq(const lbfgs_hip_vec* x, const lbfgs_hip_vec* g, int out_slot) {
    if (!same_ctx(x, g) || !slot_ok(out_slot, 2)) return LBFGS_HIP_ERR_ARG;
    OpNorms2 op{};
    op.in[0] = x->p; op.in[1] = g->p;
    double* outs[2] = {x->ctx->board + out_slot, x->ctx->board + out_slot + 1};
    return launch(x->ctx, LBFGS_HIP_K_BLAS1, op, outs);
}

// ---- history ---------------------------------------------------------------------------------
int lbfgs_hip_history_create(lbfgs_hip_ctx* ctx, int m, lbfgs_hip_history** out) {
    if (!ctx || !out || m < 1 || m > 64) return LBFGS_HIP_ERR_ARG;
    *out = nullptr;
    lbfgs_hip_history* h = new (std::nothrow) lbfgs_hip_history();
    if (!h) return fail(ctx, LBFGS_HIP_ERR_NOMEM, "out of host memory");
    h->ctx = ctx;
    h->m = m;
    h->ys = h->alpha = nullptr;
    int rc = LBFGS_HIP_OK;
    for (int i = 0; i < m && rc == LBFGS_HIP_OK; ++i) {
        lbfgs_hip_vec *s = nullptr, *y = nullptr;
        rc = lbfgs_hip_vec_alloc(ctx, &s);
        if (rc == LBFGS_HIP_OK) {
            h->s.push_back(s);
            rc = lbfgs_hip_vec_alloc(ctx, &y);
            if (rc == LBFGS_HIP_OK) h->y.push_back(y);
        }
    }
    if (rc == LBFGS_HIP_OK) {
        hipError_t e = hipMalloc(&h->ys, 2 * (size_t)m * sizeof(double));
        if (e != hipSuccess) rc = fail(ctx, LBFGS_HIP_ERR_NOMEM, "hipMalloc history scalars: %s", hipGetErrorString(e));
        else {
            h->alpha = h->ys + m;
            e = hipMemsetAsync(h->ys, 0, 2 * (size_t)m * sizeof(double), ctx->stream);
            if (e != hipSuccess) rc = fail(ctx, LBFGS_HIP_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
    }
    if (rc != LBFGS_HIP_OK) {
        lbfgs_hip_history_destroy(h);
        return rc;
    }
    *out = h;
    return LBFGS_HIP_OK;
}

void lbfgs_hip_history_destroy(lbfgs_hip_history* h) {
    if (!h) return;
    for (auto* v : h->s) lbfgs_hip_vec_free(v);
    for (auto* v : h->y) lbfgs_hip_vec_free(v);
    if (h->ys) {
        (void)hipStreamSynchronize(h->ctx->stream);
        (void)hipFree(h->ys);
    }
    if (h->gram) (void)hipFree(h->gram);
    delete h;
}

lbfgs_hip_vec* lbfgs_hip_history_s(lbfgs_hip_history* h, int slot) {
    return (h && slot >= 0 && slot < h->m) ? h->s[slot] : nullptr;
}
lbfgs_hip_vec* lbfgs_hip_history_y(lbfgs_hip_history* h, int slot) {
    return (h && slot >= 0 && slot < h->m) ? h->y[slot] : nullptr;
}

int lbfgs_hip_history_scalars_read(lbfgs_hip_history* h, double* ys, double* alpha) {
    if (!h) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    if (2 * h->m > LBFGS_HIP_BOARD_SLOTS) return LBFGS_HIP_ERR_ARG;
    const int rc_s = settle(ctx);  // (alpha is an output of the two-loop: see lbfgs_hip_vec_download)
    if (rc_s != LBFGS_HIP_OK) return rc_s;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned, h->ys, 2 * (size_t)h->m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ys) memcpy(ys, ctx->pinned, h->m * sizeof(double));
    if (alpha) memcpy(alpha, ctx->pinned + h->m, h->m * sizeof(double));
    return LBFGS_HIP_OK;
}

int lbfgs_hip_history_scalars_write(lbfgs_hip_history* h, const double* ys, const double* alpha) {
    if (!h) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    ctx->last_res.valid = false;  // (ys is an input of the latest resident two-loop)
    if (ys) HIP_TRY(ctx, hipMemcpyAsync(h->ys, ys, h->m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (alpha) HIP_TRY(ctx, hipMemcpyAsync(h->alpha, alpha, h->m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LBFGS_HIP_OK;
}

int lbfgs_hip_history_update(lbfgs_hip_history* h, int slot, const lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                             const lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, double step, int damping,
                             int out_slot) {
    if (!h || slot < 0 || slot >= h->m || !x || !xp || !g || !gp) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    if (x->ctx != ctx || xp->ctx != ctx || g->ctx != ctx || gp->ctx != ctx) return LBFGS_HIP_ERR_ARG;
    if (!slot_ok(out_slot, 7)) return LBFGS_HIP_ERR_ARG;
    double* b = ctx->board + out_slot;
    double* outs[7] = {b, b + 1, b + 2, b + 3, b + 4, b + 5, b + 6};
    // totals final inside the kernel? (one rank, or in-kernel P2P exchange)
    const bool single = ctx->comm_kind == LBFGS_HIP_COMM_NONE || ctx->comm_kind == LBFGS_HIP_COMM_P2P;
    int rc;
    if (damping) {
        OpHistUpdate<true> op{};
        op.in[0] = x->p; op.in[1] = xp->p; op.in[2] = g->p; op.in[3] = gp->p;
        op.out[0] = h->s[slot]->p; op.out[1] = h->y[slot]->p;
        op.neg_step = -step;
        rc = launch(ctx, LBFGS_HIP_K_UPDATE, op, outs, single ? h->ys + slot : nullptr, 1);
    } else {
        OpHistUpdate<false> op{};
        op.in[0] = x->p; op.in[1] = xp->p; op.in[2] = g->p; op.in[3] = gp->p;
        op.out[0] = h->s[slot]->p; op.out[1] = h->y[slot]->p;
        op.neg_step = 0.0;
        rc = launch(ctx, LBFGS_HIP_K_UPDATE, op, outs, single ? h->ys + slot : nullptr, 1);
    }
    if (rc != LBFGS_HIP_OK) return rc;
    // lbfgs.rs:656 self.ys = ys (the GLOBAL y.s): stored by the kernel itself when its totals are final, else
    // copied after the all-reduce
    if (!single)
        HIP_TRY(ctx, hipMemcpyAsync(h->ys + slot, b + 1, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return LBFGS_HIP_OK;
}

int lbfgs_hip_history_damp(lbfgs_hip_history* h, int slot, const lbfgs_hip_vec* gp, double step, double theta) {
    if (!h || slot < 0 || slot >= h->m || !gp || gp->ctx != h->ctx) return LBFGS_HIP_ERR_ARG;
    OpDamp op{};
    op.in[0] = gp->p; op.in[1] = h->y[slot]->p; op.out[0] = h->y[slot]->p;
    op.neg_step = -step;
    op.one_minus_theta = 1.0 - theta;
    op.theta = theta;
    return launch(h->ctx, LBFGS_HIP_K_UPDATE, op, nullptr);
}

// ---- two-loop recursion ------------------------------------------------------------------------

int lbfgs_hip_two_loop(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                       int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int* new_end) {
    return lbfgs_hip_two_loop_from(h, d, g, k, end, gamma_num_slot, gamma_den_slot, dnorm_slot, -1, new_end);
}

int lbfgs_hip_two_loop_from(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                            int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end) {
    return two_loop_impl(h, d, g, k, end, gamma_num_slot, gamma_den_slot, dnorm_slot, first_dot_slot, new_end, false, 0, 0);
}

int lbfgs_hip_two_loop_owlqn(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* pg, uint64_t k, int end,
                             int gamma_num_slot, int gamma_den_slot, int dnorm_slot, uint64_t start, uint64_t end_,
                             int* new_end) {
    if (!slot_ok(dnorm_slot, 4)) return LBFGS_HIP_ERR_ARG;
    return two_loop_impl(h, d, pg, k, end, gamma_num_slot, gamma_den_slot, dnorm_slot, -1, new_end, true, start, end_);
}

}  // extern "C"

// ---- the two-loop as ONE kernel with the running vector resident in registers + LDS (resident.h) ----------------
namespace {
#if LH_RES_TRACE
// A/B builds only (tools/handoff_trace.sh): every workgroup leaves wall-clock stamps (100 MHz, one clock for the whole chip)
// of every hand-off in the ticket form's partials buffer -- arrival, partial published, totals known.  From those: how far
// apart the workgroups ARRIVE (skew: load imbalance of the step, not a property of the hand-off), how long the hand-off
// takes once the LAST workgroup has arrived (its true latency), and what a workgroup sees on average.
struct ResTraceAcc {
    double skew = 0, after_last = 0, whole_mean = 0, sums = 0, step = 0, polls = 0, spread_out = 0;
    unsigned long long handoffs = 0, steps = 0;
    // per workgroup: how late it arrives (ticks after the FIRST arrival of the same hand-off), how often it is the last / the
    // first one in, where it ran (XCC_ID << 16 | HW_ID[15:0]) and on how many launches that differed from the launch before
    double late[256] = {0}, late_sq[256] = {0};
    unsigned long long last_cnt[256] = {0}, first_cnt[256] = {0}, hw_moves[256] = {0};
    unsigned int hw[256] = {0};
    bool hw_seen = false;
    int grid = 0;
    double skew_by_handoff[64] = {0};  // by position within the launch (0 = the first hand-off of a launch)
    unsigned long long n_by_handoff[64] = {0};
} g_res_trace;
void res_trace_collect(lbfgs_hip_ctx* ctx, unsigned int first_tag, int handoffs, int grid) {
    static std::vector<unsigned long long> host(64 * 256 * 4);
    if (handoffs > 60 || grid > 256 || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(host.data(), ctx->partials, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return;
    long long prev_out = 0;
    ResTraceAcc& a = g_res_trace;
    a.grid = grid;
    for (int h = 0; h < handoffs; ++h) {
        const unsigned long long* tr = host.data() + (size_t)((first_tag + (unsigned int)h) & 63u) * 256 * 4;
        long long in_min = 0, in_max = 0, out_min = 0, out_max = 0;
        int w_first = 0, w_last = 0;
        double whole = 0, sums = 0, polls = 0;
        for (int w = 0; w < grid; ++w) {
            const long long t0 = (long long)tr[w * 4], t1 = (long long)tr[w * 4 + 1], t2 = (long long)tr[w * 4 + 2];
            if (w == 0 || t0 < in_min) { in_min = t0; w_first = w; }
            if (w == 0 || t0 > in_max) { in_max = t0; w_last = w; }
            if (w == 0 || t2 < out_min) out_min = t2;
            if (w == 0 || t2 > out_max) out_max = t2;
            whole += (double)(t2 - t0);
            sums += (double)(t1 - t0);
            polls += (double)(tr[w * 4 + 3] & 0xffffffffull);
        }
        for (int w = 0; w < grid; ++w) {
            const double l = (double)((long long)tr[w * 4] - in_min);
            a.late[w] += l;
            a.late_sq[w] += l * l;
        }
        a.last_cnt[w_last] += 1;
        a.first_cnt[w_first] += 1;
        a.skew_by_handoff[h] += (double)(in_max - in_min);
        a.n_by_handoff[h] += 1;
        a.skew += (double)(in_max - in_min);
        a.after_last += (double)(out_min - in_max);
        a.spread_out += (double)(out_max - out_min);
        a.whole_mean += whole / grid;
        a.sums += sums / grid;
        a.polls += polls / grid;
        a.handoffs += 1;
        if (h > 0) { a.step += (double)(in_min - prev_out); a.steps += 1; }
        prev_out = out_max;
    }
    {   // placement of the workgroups in this launch (any hand-off's stamps carry it)
        const unsigned long long* tr = host.data() + (size_t)(first_tag & 63u) * 256 * 4;
        for (int w = 0; w < grid; ++w) {
            const unsigned int hw = (unsigned int)(tr[w * 4 + 3] >> 32) & 0xfff00u;  // xcc, se, sh, cu (not pipe / simd / wave)
            if (a.hw_seen && hw != a.hw[w]) a.hw_moves[w] += 1;
            a.hw[w] = hw;
        }
        a.hw_seen = true;
    }
}
}  // namespace
void res_trace_print() {
    const ResTraceAcc& a = g_res_trace;
    if (!a.handoffs) return;
    const double k = 10.0 / (double)a.handoffs;  // ticks of 10 ns -> ns per hand-off
    fprintf(stderr, "[res-trace] %llu hand-offs: a workgroup spends %.0f ns in one (sums+publish %.0f, %.2f polls); workgroups ARRIVE "
                    "%.0f ns apart; first one out %.0f ns after the LAST arrival, last one out %.0f ns later; "
                    "streaming between hand-offs (last out -> first in) %.0f ns\n", a.handoffs, a.whole_mean * k, a.sums * k,
            a.polls / (double)a.handoffs, a.skew * k, a.after_last * k, a.spread_out * k, a.steps ? a.step * 10.0 / (double)a.steps : 0.0);
    // ---- is the arrival skew SYSTEMATIC (the same workgroups / XCDs / CUs late every time) or random?
    const int G = a.grid;
    const double H = (double)a.handoffs;
    double mean_all = 0, between = 0, within = 0;
    for (int w = 0; w < G; ++w) mean_all += a.late[w] / H;
    mean_all /= G;
    for (int w = 0; w < G; ++w) {
        const double mw = a.late[w] / H;
        between += (mw - mean_all) * (mw - mean_all);
        within += a.late_sq[w] / H - mw * mw;
    }
    between /= G; within /= G;
    unsigned long long moves = 0;
    for (int w = 0; w < G; ++w) moves += a.hw_moves[w];
    fprintf(stderr, "[res-skew] grid %d: mean lateness (after the first arrival) %.0f ns; variance between workgroups %.3g ns^2, within a "
                    "workgroup (hand-off to hand-off) %.3g ns^2 => %.0f %% of the variance is systematic per workgroup; workgroup->CU "
                    "placement changed %llu times between consecutive launches (of %d workgroups x launches)\n", G, mean_all * 10.0,
            between * 100.0, within * 100.0, 100.0 * between / std::max(between + within, 1e-300), moves, G);
    // by XCC (8 on MI355X) and by shader engine within it
    double xl[16] = {0}; unsigned long long xlast[16] = {0}; int xn[16] = {0};
    for (int w = 0; w < G; ++w) {
        const int x = (a.hw[w] >> 16) & 15;
        xl[x] += a.late[w] / H; xlast[x] += a.last_cnt[w]; xn[x] += 1;
    }
    for (int x = 0; x < 16; ++x)
        if (xn[x])
            fprintf(stderr, "[res-skew] xcc %d: %3d workgroups, mean lateness %5.0f ns, last one in %5.1f %% of the hand-offs\n", x, xn[x],
                    xl[x] / xn[x] * 10.0, 100.0 * (double)xlast[x] / H);
    {   // by position within the launch
        char buf[1024]; int o = 0;
        for (int h = 0; h < 64 && a.n_by_handoff[h]; ++h)
            o += snprintf(buf + o, sizeof(buf) - (size_t)o, " %.0f", a.skew_by_handoff[h] * 10.0 / (double)a.n_by_handoff[h]);
        fprintf(stderr, "[res-skew] skew by hand-off position within the launch (ns):%s\n", buf);
    }
    // the 12 latest and the 12 earliest workgroups
    std::vector<int> order(G);
    for (int w = 0; w < G; ++w) order[w] = w;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return a.late[x] > a.late[y]; });
    auto row = [&](int w) {
        const double mw = a.late[w] / H, sd = std::sqrt(std::max(0.0, a.late_sq[w] / H - mw * mw));
        fprintf(stderr, "[res-skew]   wg %3d  xcc %u se %u sh %u cu %2u  late %5.0f +- %4.0f ns  last %5.1f %%  first %5.1f %%\n", w,
                (a.hw[w] >> 16) & 15, (a.hw[w] >> 13) & 7, (a.hw[w] >> 12) & 1, (a.hw[w] >> 8) & 15, mw * 10.0, sd * 10.0,
                100.0 * (double)a.last_cnt[w] / H, 100.0 * (double)a.first_cnt[w] / H);
    };
    for (int i = 0; i < std::min(G, 12); ++i) row(order[i]);
    fprintf(stderr, "[res-skew]   ...\n");
    for (int i = std::max(0, G - 12); i < G; ++i) row(order[i]);
    if (const char* path = getenv("LBFGS_HIP_TRACE_FILE")) {  // every workgroup, for offline reading
        if (FILE* f = fopen(path, "a")) {
            fprintf(f, "# grid %d handoffs %llu\nwg,xcc,se,sh,cu,late_ns,late_sd_ns,last_frac,first_frac,placement_moves\n", G, a.handoffs);
            for (int w = 0; w < G; ++w) {
                const double mw = a.late[w] / H, sd = std::sqrt(std::max(0.0, a.late_sq[w] / H - mw * mw));
                fprintf(f, "%d,%u,%u,%u,%u,%.0f,%.0f,%.4f,%.4f,%llu\n", w, (a.hw[w] >> 16) & 15, (a.hw[w] >> 13) & 7, (a.hw[w] >> 12) & 1,
                        (a.hw[w] >> 8) & 15, mw * 10.0, sd * 10.0, (double)a.last_cnt[w] / H, (double)a.first_cnt[w] / H, a.hw_moves[w]);
            }
            fclose(f);
        }
    }
}
namespace {
#endif
template <int ER, bool HYB = false>
int resident_launch(lbfgs_hip_ctx* ctx, const ResArgs& ra, const RedCtl& red, int grid, size_t lds_bytes, bool nt) {
    auto kern_nt = two_loop_resident_kernel<ER, true, HYB>;
    auto kern_pl = two_loop_resident_kernel<ER, false, HYB>;
    // more than 64 KiB of dynamic LDS has to be asked for, once per kernel AND device: remembered in the context
    const unsigned long long bit = 1ull << ((ER / 8 + (HYB ? 8 : 0)) * 2 + (nt ? 1 : 0));  // ER/8 in {0, 1, 3, 5, 7}: bits 0..31
    if (!(ctx->resident_attr_mask & bit)) {
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(nt ? kern_nt : kern_pl),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_PAIRS_MAX * BLOCK * (int)sizeof(d2)));
        ctx->resident_attr_mask |= bit;
    }
    if (nt) hipLaunchKernelGGL(kern_nt, dim3(grid), dim3(BLOCK), lds_bytes, ctx->stream, ra, red);
    else hipLaunchKernelGGL(kern_pl, dim3(grid), dim3(BLOCK), lds_bytes, ctx->stream, ra, red);
    HIP_TRY(ctx, hipGetLastError());
    return LBFGS_HIP_OK;
}

// (stream.h describes it; anonymous namespace: this translation unit only)
__global__ __launch_bounds__(64) void ext_post_gate_kernel(const ExtGateArgs a) {
    if (threadIdx.x != 0) return;
    if (a.post_epoch) __hip_atomic_store(a.flags + 1, a.post_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.gate_epoch) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.gate_epoch) {
            if (__hip_atomic_load(a.flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= a.abort_id) break;
            if ((unsigned long long)(wall_clock64() - t0) > a.timeout_ticks) {
                atomicExch(a.err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// The GATED exchange's second stream (stream.h): for each of the launch's `handoffs` exchanges  gate(epoch) -> ncclAllReduce of
// the slot -> post(epoch), the post of one exchange and the gate of the next in ONE kernel.  Enqueued right behind the launch of
// the persistent kernel: every element is in the stream long before the kernel reaches the hand-off it serves (a step takes
// tens of microseconds, an element a few to enqueue); a pre-enqueued round trip through two kernel boundaries costs 3.9 us
// on one GPU (tools/gated_exchange_probe.hip) + whatever the all-reduce itself takes.
// Epochs: DevCounters::p2p_epoch as the kernel will walk it -- exchange number c (counted from 0 over the context's life) has
// epoch c % (2^32 - 1) + 1, never 0.  Every hand-off carries one sum except the last (2, or 4 under OWL-QN).
int enqueue_gated_chain(lbfgs_hip_ctx* ctx, unsigned long long first_exchange, int handoffs, int last_count) {
    auto epoch_of = [](unsigned long long c) { return (unsigned long long)(c % 0xFFFFFFFFull) + 1ull; };
    double* const ring = reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->ext_block) + EXT_BUF_OFFSET);
    ctx->ext_launches += 1;
    ExtGateArgs ga{};
    ga.flags = ctx->ext_block;
    ga.err = ctx->p2p_err;
    ga.abort_id = ctx->ext_launches;
    ga.timeout_ticks = ctx->p2p_timeout_ticks + ctx->handoff_timeout_ticks;  // (the kernel may itself be waiting for its workgroups)
    ga.post_epoch = 0;
    // the first gate starts its clock when the persistent kernel is about to start, not while that kernel is still queued behind
    // earlier work of the compute stream (which may be the caller's own stream): two_loop_resident recorded ext_ev just before it
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->xstream, ctx->ext_ev, 0));
    for (int i = 0; i < handoffs; ++i) {
        const unsigned long long e = epoch_of(first_exchange + (unsigned long long)i);
        ga.gate_epoch = e;
        hipLaunchKernelGGL(ext_post_gate_kernel, dim3(1), dim3(64), 0, ctx->xstream, ga);
        const int rc = rccl_allreduce_on(ctx, ring + (size_t)(e % (unsigned long long)EXT_SLOTS) * EXT_SLOT_DOUBLES,
                                         i + 1 == handoffs ? last_count : 1, ctx->xstream);
        if (rc != LBFGS_HIP_OK) {
            ext_abort(ctx);
            return rc;
        }
        ga.post_epoch = e;
        ctx->allreduce_calls += 1;
    }
    ga.gate_epoch = 0;
    hipLaunchKernelGGL(ext_post_gate_kernel, dim3(1), dim3(64), 0, ctx->xstream, ga);
    HIP_TRY(ctx, hipGetLastError());
    return LBFGS_HIP_OK;
}

// The gated exchange PROVEN on this context's own communicator before a two-loop relies on it (lbfgs_hip_ctx_create ->
// ext_prepare; collective).  What cannot be known without trying: that RCCL's all-reduce kernel is dispatched, finds compute
// units and reaches its peers while a kernel of one workgroup per CU (the whole LDS each, like the resident two-loop) sits on
// all but eight CUs waiting for it.  So exactly that runs once: workgroup 0 hands (rank + 1) to ext_exchange, the chain
// gate -> ncclAllReduce -> post is enqueued on the second stream, and the sum must come back as world*(world+1)/2 within a
// second.  The other workgroups wait for workgroup 0's word, bounded.  A rank that times out still lets its chain run to the
// end (the gate has seen its flag), so no peer is left waiting inside RCCL; the ranks then agree on the outcome (ext_prepare).
__global__ __launch_bounds__(BLOCK) void ext_selftest_resident_kernel(const P2PCtl c, const unsigned int epoch, const double mine) {
    extern __shared__ char whole_cu[];  // (a CU's LDS: nothing else shares the CU)
    (void)whole_cu;
    __shared__ double vals[EXT_SLOT_DOUBLES];
    unsigned long long* const done = c.ext_flags + 5;
    if (blockIdx.x != 0) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)epoch &&
                   (unsigned long long)(wall_clock64() - t0) < 2ull * c.timeout_ticks)
                __builtin_amdgcn_s_sleep(8);
        }
        return;
    }
    if (threadIdx.x == 0) vals[0] = mine;
    __syncthreads();
    ext_exchange(c, epoch, vals, 1);
    if (threadIdx.x == 0) {
        __hip_atomic_store(c.ext_flags + 6, (unsigned long long)__double_as_longlong(vals[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(done, (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace
int ext_selftest_resident(lbfgs_hip_ctx* ctx, bool* ok, bool* enqueued) {
    *ok = false;
    *enqueued = false;  // (has this rank's all-reduce of the trial been handed to RCCL?  Its peers wait for it: ext_prepare)
    constexpr unsigned long long EPOCH = 0x7E57ull;         // (any non-zero epoch: the words are cleared again below)
    constexpr unsigned long long TICKS = 100000000ull;      // 1 s of the 100 MHz wall clock
    unsigned long long* const w = ctx->ext_block;           // words: 0 A, 1 B, 2 abort, (3, 4: the stream handshake), 5 done, 6 sum, 7 err
    double* const ring = reinterpret_cast<double*>(reinterpret_cast<char*>(w) + EXT_BUF_OFFSET);
    double* const slot = ring + (size_t)(EPOCH % (unsigned long long)EXT_SLOTS) * EXT_SLOT_DOUBLES;
    P2PCtl c{};
    c.world = ctx->shard.world;
    c.rank = ctx->shard.rank;
    c.err = reinterpret_cast<unsigned int*>(w + 7);
    c.timeout_ticks = TICKS;
    c.ext_buf = ring;
    c.ext_flags = w;
    const size_t lds_bytes = (size_t)RES_LDS_PAIRS_MAX * BLOCK * sizeof(d2);  // (what the resident two-loop asks for)
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ext_selftest_resident_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const int grid = std::max(1, ctx->cu_count - 8);  // (two_loop_resident's grid_cap under the gated exchange)
    hipLaunchKernelGGL(ext_selftest_resident_kernel, dim3(grid), dim3(BLOCK), lds_bytes, ctx->stream, c, (unsigned int)EPOCH,
                       (double)(ctx->shard.rank + 1));
    ctx->ext_launches += 1;
    ExtGateArgs ga{};
    ga.flags = w;
    ga.err = c.err;
    ga.abort_id = ctx->ext_launches;
    ga.timeout_ticks = 3ull * TICKS;
    ga.gate_epoch = EPOCH;
    hipLaunchKernelGGL(ext_post_gate_kernel, dim3(1), dim3(64), 0, ctx->xstream, ga);
    const int rc = rccl_allreduce_on(ctx, slot, 1, ctx->xstream);
    *enqueued = true;
    ga.post_epoch = EPOCH;
    ga.gate_epoch = 0;
    hipLaunchKernelGGL(ext_post_gate_kernel, dim3(1), dim3(64), 0, ctx->xstream, ga);  // (also after a failed enqueue: the kernel is let go)
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->xstream));
    if (rc != LBFGS_HIP_OK) return rc;
    unsigned long long seen[3] = {0, 0, 0};
    HIP_TRY(ctx, hipMemcpy(seen, w + 5, sizeof(seen), hipMemcpyDeviceToHost));
    double sum = 0.0;
    memcpy(&sum, &seen[1], sizeof(sum));
    const double W = (double)ctx->shard.world;
    *ok = seen[0] == EPOCH && (seen[2] & 0xFFFFFFFFull) == 0ull && sum == 0.5 * W * (W + 1.0);
    // back to the state the first launch expects: flags, handshake and self-test words and the slot cleared (the abort word stays)
    HIP_TRY(ctx, hipMemsetAsync(w, 0, 2 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(w + 3, 0, 5 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(slot, 0, EXT_SLOT_DOUBLES * sizeof(double), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LBFGS_HIP_OK;
}
namespace {

// What two_loop_resident decides from the SHARD and the DEVICE alone (nothing of the call: bound and the tag wrap are the same
// on every rank).  Under the gated RCCL exchange every rank must take the same launch form -- a rank with a kernel per step
// next to ranks with gated chains would pair its all-reduces wrongly -- so lbfgs_hip_ctx_create evaluates exactly this on every
// rank and the ranks agree on it (resident_shard_eligible, context.hip ext_prepare).
constexpr int RES_ER_MAX = 60;
struct ResPlan {
    int grid, er;            // workgroups; register rounds of the instantiation
    uint64_t per_round, E;   // 16-byte pairs per round of the grid; rounds (pairs per thread)
    bool hybrid;             // the shard does not fit the chip: rounds beyond er + el stay in HBM
    uint32_t eh, el;         // rounds whose q stays in HBM; rounds in LDS
};
bool resident_plan(const lbfgs_hip_ctx* ctx, bool gated, ResPlan* pl) {
    // one workgroup per CU: all of them resident at once (fewer on request -- tests run two ranks on one GPU -- and for
    // vectors of a few MB, where a hand-off among fewer workgroups is worth more than the idle CUs' bandwidth: >= 8 pairs per
    // thread, at least 64 workgroups; measured at n = 1e5 / 3e5 / 1e6, profiles/r02_resident_small_n.log)
    const uint64_t n = ctx->shard.n_local;
    // ... and 27/32 of the CUs, the streaming kernels' grid, once less than a fifth of a shard fits the chip and
    // the kernel is a streaming kernel above all (n = 1e8: 9.73 ms against 9.83 with 256; at 2.5e7 256 wins by 4 %)
    const bool mostly_streaming = (n >> 1) > 5ull * (uint64_t)(RES_ER_MAX + RES_LDS_PAIRS_MAX) * (uint64_t)ctx->cu_count * BLOCK;
    const int grid_auto = mostly_streaming ? std::max(1, ctx->cu_count * 27 / 32)
        : (int)std::min<uint64_t>((uint64_t)ctx->cu_count, std::max<uint64_t>(64, ((n >> 1) + BLOCK * 8 - 1) / (BLOCK * 8)));
    // (gated: RCCL's kernel and the gates need CUs of their own while this kernel waits -- one per XCD is left free, workgroups
    // being dealt to the XCDs round robin; tools/gated_exchange_probe.hip measured the same round trip with 1, 8 or 16 free)
    // (leaving ONE free so that rank 0's shard of the metric's 8-GPU run -- 12 500 224 elements = 255 workgroups of 96 rounds -- stays
    // all on the chip was measured and is no better: 783 us against 755 us for the hybrid launch on 248, profiles/r05_rccl_one_rank.log)
    const int grid_cap = gated ? std::max(1, ctx->cu_count - 8) : ctx->cu_count;
    const int grid = std::min(ctx->resident_grid > 0 ? std::min(ctx->resident_grid, ctx->cu_count) : grid_auto, grid_cap);
    if (grid < 1 || grid > BLOCK || grid > MAX_GRID) return false;  // (every thread polls one workgroup's granules)
    const uint64_t per_round = (uint64_t)grid * BLOCK;
    const uint64_t E = ((n >> 1) + per_round - 1) / per_round;  // 16-byte pairs per thread
    // Larger shards: HYBRID -- the first 60 + 36 rounds of every thread stay on the chip, the rest of q lives in `d` and is
    // streamed by every step as on the kernel-per-step path (resident.h).
    const bool hybrid = E > (uint64_t)(RES_ER_MAX + RES_LDS_PAIRS_MAX);
    if (hybrid && !ctx->resident_hybrid) return false;
    if (E == 0 || (n >> 1) + per_round * 4 >= (1ull << 28)) return false;
    // Rounds 0 .. E-2 are full for every thread, round E-1 is the ragged one.  The register rounds carry no bounds checks,
    // so ER <= E-1; the rest (the ragged round included) lives in LDS -- everything, for the smallest vectors (ER = 0).
    const int er = E - 1 >= RES_ER_MAX ? RES_ER_MAX : E - 1 >= 40 ? 40 : E - 1 >= 24 ? 24 : E - 1 >= 8 ? 8 : 0;
    const uint32_t eh = hybrid ? (uint32_t)(E - RES_ER_MAX - RES_LDS_PAIRS_MAX) : 0u;  // rounds whose q stays in HBM
    const uint32_t el = hybrid ? (uint32_t)RES_LDS_PAIRS_MAX : (uint32_t)((E - er + RES_UNROLL - 1) / RES_UNROLL * RES_UNROLL);
    if (el > (uint32_t)RES_LDS_PAIRS_MAX) return false;
    *pl = ResPlan{grid, er, per_round, E, hybrid, eh, el};
    return true;
}
// once per context: can this device hold one such workgroup per CU, and does this queue reach every CU?
bool resident_device_ok(lbfgs_hip_ctx* ctx) {
    constexpr int ER_MAX = RES_ER_MAX;
    if (ctx->resident_ok < 0) {  // once: can this device hold one such workgroup per CU?
        int nb = 0;
        const size_t lds_max = (size_t)RES_LDS_PAIRS_MAX * BLOCK * sizeof(d2);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(two_loop_resident_kernel<ER_MAX, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e == hipSuccess)
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, two_loop_resident_kernel<ER_MAX, false, true>, BLOCK, lds_max);
        ctx->resident_ok = (e == hipSuccess && nb >= 1) ? 1 : 0;
        (void)hipGetLastError();
        // ... and does this queue reach every CU the device reports?  A CU-masked stream (hipExtStreamCreateWithCUMask,
        // HSA_CU_MASK / ROC_GLOBAL_CU_MASK) would leave part of the grid waiting for CUs it never gets: kernel per step then.
        // (A partitioned device -- CPX / DPX -- reports its own CU count, so it needs no special case.)
        if (ctx->resident_ok == 1) {
            uint32_t mask[32] = {0};
            if (hipExtStreamGetCUMask(ctx->stream, 32, mask) == hipSuccess) {
                int cus = 0;
                for (uint32_t w : mask) cus += __builtin_popcount(w);
                if (cus > 0 && cus < ctx->cu_count) {
                    ctx->resident_ok = 0;
                    fprintf(stderr, "[lbfgs_hip] note: this stream reaches %d of the device's %d CUs (CU mask): the two-loop runs with a "
                                    "kernel per step instead of the chip-wide resident kernel\n", cus, ctx->cu_count);
                }
            }
            (void)hipGetLastError();
            const char* gm = getenv("ROC_GLOBAL_CU_MASK");
            const char* hm = getenv("HSA_CU_MASK");
            if ((gm && *gm) || (hm && *hm)) {
                ctx->resident_ok = 0;
                fprintf(stderr, "[lbfgs_hip] note: a global CU mask is set in the environment: the two-loop runs with a kernel per step\n");
            }
        }
        if (getenv("LBFGS_HIP_VERBOSE"))
            fprintf(stderr, "[lbfgs_hip] resident two-loop kernel: %s (%s, %d workgroup(s) per CU with %zu bytes of LDS)\n",
                    ctx->resident_ok ? "usable" : "not usable", hipGetErrorString(e), nb, lds_max);
    }
    return ctx->resident_ok == 1;
}
}  // namespace
// (context.hip ext_prepare: this rank's vote on the gated form)
bool resident_shard_eligible(lbfgs_hip_ctx* ctx) {
    ResPlan pl{};
    return ctx->resident_on && !ctx->handoff_ticket && ctx->grid_override == 0 && resident_plan(ctx, true, &pl) && resident_device_ok(ctx);
}
namespace {

// -> 1 if the recursion was launched as the resident kernel, 0 if this case is not eligible (caller falls back), < 0 error
int two_loop_resident(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                      int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end, bool owl,
                      uint64_t owl_start, uint64_t owl_end) {
    lbfgs_hip_ctx* ctx = h->ctx;
    const int m = h->m;
    const int bound = (int)std::min<uint64_t>((uint64_t)m, k);
    // One rank -- or several that each have their GPU to themselves: a kernel that fills the chip and waits for its peers
    // inside would starve ranks sharing the GPU (lbfgs_hip_comm.exclusive_device).
    // Under RCCL the exchange is GATED (stream.h ext_exchange): ncclAllReduce only exists as a host-enqueued kernel, so the host
    // enqueues one per hand-off on a second stream, each behind a gate kernel that waits for this kernel's flag -- q stays on the chip.
    const bool gated = ctx->comm_kind == LBFGS_HIP_COMM_RCCL && ctx->rccl_resident && ctx->p2p_exclusive && ctx->xstream && ctx->ext_block && ctx->ext_ev;
    const bool comm_ok = ctx->comm_kind == LBFGS_HIP_COMM_NONE || (ctx->comm_kind == LBFGS_HIP_COMM_P2P && ctx->p2p_exclusive) || gated;
    if (!ctx->resident_on || !comm_ok || ctx->handoff_ticket || bound < 1 || ctx->grid_override > 0 ||
        2 * bound > RES_MAX_STEPS)
        return 0;
    // (gated: lbfgs_hip_ctx_create made the second stream and the uncached block, proved that both streams are served concurrently
    // -- else rccl_resident is off --, let RCCL run its first collective on that stream, and the ranks have AGREED that every
    // shard and every device can take this form: resident_shard_eligible below, context.hip ext_prepare)
    ResPlan pl{};
    if (!resident_plan(ctx, gated, &pl)) return 0;
    if (ctx->red_count % 0xFFFFFFFFull + 2ull * (uint64_t)bound + 4ull >= 0xFFFFFFFFull) return 0;  // tag wrap: eager path
    if (!resident_device_ok(ctx)) return 0;
    const uint64_t n = ctx->shard.n_local;
    const int grid = pl.grid, er = pl.er;
    const uint64_t per_round = pl.per_round, E = pl.E;
    const bool hybrid = pl.hybrid;
    const uint32_t eh = pl.eh, el = pl.el;
    constexpr int ER_MAX = RES_ER_MAX;

    ProfScope whole(ctx, LBFGS_HIP_K_TWOLOOP_ALL);
    const int e1 = (end + 1) % m;                                   // lbfgs.rs:577
    auto jat = [&](int i) { return ((e1 - 1 - i) % m + m) % m; };   // lbfgs.rs:583
    ResArgs ra{};
    ra.g = g->p;
    ra.d = d->p;
    ra.first_dot = first_dot_slot >= 0 ? ctx->board + first_dot_slot : nullptr;
    ra.first_s = h->s[jat(0)]->p;
    ra.gnum = ctx->board + gamma_num_slot;
    ra.gden = ctx->board + gamma_den_slot;
    ra.out_dn = ctx->board + dnorm_slot;
    ra.n = n;
    ra.gofs = ctx->shard.offset;
    ra.owl = owl ? 1 : 0;
    ra.owl_start = owl_start;
    ra.owl_end = owl_end;
    ra.pairs_per_thread = (uint32_t)E;
    ra.lds_pairs = el;
    ra.hbm_pairs = eh;
    // the slice of the HBM part of q that is kept out of the `nt` stream (resident.h res_hbm_rounds): so many MiB of it
    ra.hbm_plain_pairs = eh ? (uint32_t)std::min<uint64_t>(eh, (ctx->resident_plain_bytes + per_round * 16 - 1) / (per_round * 16)) : 0u;
    if (ctx->resident_plain_bytes == 0) ra.hbm_plain_pairs = 0;
    // rounds a waiting workgroup touches behind the window (resident.h TOUCHING): 16 of two vectors fill an XCD's L2, which
    // pays from ~64 rounds per thread up; shorter steps want 8 (profiles/r04_touch_sweep.log)
    // (the kernels are instantiated with exactly these depths: resident.h, TOUCH)
    ra.touch_rounds = (uint32_t)(ctx->resident_touch >= 0 ? ctx->resident_touch : (er == 60 ? 16 : 8));
    int ns = 0;
    auto add = [&](const double* u, const double* v, int j, int mode_b, int scale, int aidx, int last) {
        ResStep& st = ra.step[ns++];
        st.u = u; st.v = v; st.ys = h->ys + j; st.alpha = h->alpha + j;
        st.mode_b = mode_b; st.scale = scale; st.alpha_idx = aidx; st.last = last;
    };
    for (int i = 1; i < bound; ++i)   // first loop: q -= alpha_{i-1} y_{j_{i-1}} ; next numerator s_{j_i} . q
        add(h->y[jat(i - 1)]->p, h->s[jat(i)]->p, jat(i - 1), 0, 0, i - 1, 0);
    add(h->y[jat(bound - 1)]->p, nullptr, jat(bound - 1), 0, 1, bound - 1, 0);  // transition: *gamma ; y_last . q
    for (int i = bound - 1; i >= 1; --i)  // second loop: q += (alpha_j - beta_j) s_j ; next numerator y_{j-1} . q
        add(h->s[jat(i)]->p, h->y[jat(i - 1)]->p, jat(i), 1, 0, i, 0);
    add(h->s[jat(0)]->p, g->p, jat(0), 1, 0, 0, 1);  // last step: ||d||^2 and g.d
    ra.nsteps = ns;
    ra.fault_drop_last = (ctx->resident_fault > 0 && --ctx->resident_fault == 0) ? 1 : 0;  // (the k-th resident launch of the context)

    RedCtl red{};
    bool in_kernel_exchange = false;
    double* outs2[4] = {ra.out_dn, ra.out_dn + 1, ra.out_dn + 2, ra.out_dn + 3};
#if LH_RES_TRACE
    const unsigned int trace_first_tag = (unsigned int)(ctx->red_count % 0xFFFFFFFFull) + 1u;
#endif
    const int rc_p = prep_red(ctx, red, owl ? 4 : 2, outs2, nullptr, 0, &in_kernel_exchange);
    if (rc_p != LBFGS_HIP_OK) return rc_p;
    const unsigned long long first_exchange = ctx->p2p_count;  // (gated: exchanges closed so far = the epochs this launch starts behind)
    if (gated) {
        red.p2p.world = ctx->shard.world;
        red.p2p.rank = ctx->shard.rank;
        red.p2p.err = ctx->p2p_err;
        red.p2p.timeout_ticks = ctx->p2p_timeout_ticks;
        red.p2p.ext_flags = ctx->ext_block;
        red.p2p.ext_buf = reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->ext_block) + EXT_BUF_OFFSET);
        in_kernel_exchange = true;  // (the totals are final when the kernel ends; the counters below are the P2P form's)
        ctx->p2p_count += 1;
        ctx->two_loop_exchanges += 1;
        // the totals ARE final inside the kernel: mirror them like the P2P form does (prep_red decided for "not final")
        if (ctx->mirror) {
            bool any_public = false;
            for (int k = 0; k < (owl ? 4 : 2); ++k) any_public = any_public || (outs2[k] - ctx->board >= 0 && outs2[k] - ctx->board < LBFGS_HIP_BOARD_SLOTS);
            if (any_public) {
                for (int k = 0; k < (owl ? 4 : 2); ++k) {
                    const long idx = outs2[k] - ctx->board;
                    if (idx >= 0 && idx < LBFGS_HIP_BOARD_SLOTS + 2) ctx->mirror_valid[idx] = true;
                }
                red.mirror.host_board = ctx->mirror_dev;
                red.mirror.board = ctx->board;
                red.mirror.host_seq = reinterpret_cast<unsigned long long*>(ctx->mirror_dev + LBFGS_HIP_BOARD_SLOTS + 2);
                ctx->mirror_seq += 1;
                red.mirror.host_err = reinterpret_cast<unsigned long long*>(ctx->mirror_dev + LBFGS_HIP_BOARD_SLOTS + 3);
                red.mirror.slots = LBFGS_HIP_BOARD_SLOTS + 2;
            }
        }
    }
    // until a resident launch of this context has been seen to complete, a missing workgroup costs milliseconds (see
    // lbfgs_hip_ctx::first_timeout_ticks); one rank only: with peers a hand-off also waits for THEIR start-up
    if (!ctx->resident_proven && ctx->comm_kind == LBFGS_HIP_COMM_NONE)
        red.timeout_ticks = std::min(red.timeout_ticks, ctx->first_timeout_ticks);
    // Afterwards a resident hand-off still waits only HALF as long as a launch-per-step reduction does.  Two contexts sharing a
    // GPU can block each other: this kernel holds most CUs and needs the rest EMPTY, while the other stream's reducing kernel
    // has its waiting last workgroup on one of those and cannot place its remaining workgroups on the XCDs this kernel fills
    // (workgroups are bound to XCDs round robin).  Only a timeout ends that, and it must be THIS kernel's: its inputs are
    // intact and it is re-run per step (context.hip recover_resident), whereas a timed-out launch-per-step reduction is an
    // error.  Both sides start waiting within a step's time of each other; half the timeout is the margin.
    else
        red.timeout_ticks = std::max(1ull, red.timeout_ticks / 2ull);
    // (one hand-off per step -- under OWL-QN the last step's travels after the projection, with four values -- plus one
    // for the first numerator if it is summed here: the SAME sequence of reductions as the launch-per-step path, so under
    // P2P ranks whose eligibility differs -- an empty shard, a shard one round larger -- still meet in every exchange)
    const unsigned long long handoffs = (unsigned long long)(ns + (ra.first_dot ? 0 : 1));
    ctx->red_count += handoffs - 1ull;  // one tag per hand-off (prep_red counted one)
    if (in_kernel_exchange) {  // ... and one P2P epoch each (prep_red counted one)
        ctx->p2p_count += handoffs - 1ull;
        ctx->two_loop_exchanges += handoffs - 1ull;
    }
    // streaming hints on the history vectors' loads (read once per step) and on d's stores: from 16 MiB vectors up for this
    // kernel (measured: neutral at 16 MB, +2 % at 24 MB, +8 % at 32-48 MB, profiles/r02_resident_nt_threshold.log; the
    // launch-per-step kernels want them from 64 / 128 MiB only: their running vector is re-read from the caches)
    const bool nt = n * sizeof(double) >= ctx->resident_nt_bytes;
    const size_t lds_bytes = (size_t)el * BLOCK * sizeof(d2);
    int rc;
    if (gated) HIP_TRY(ctx, hipEventRecord(ctx->ext_ev, ctx->stream));  // (enqueue_gated_chain: the chain's first gate waits for it)
    {
        ProfScope ps(ctx, LBFGS_HIP_K_TWOLOOP_RESIDENT);
        switch (er) {
            case 0: rc = resident_launch<0>(ctx, ra, red, grid, lds_bytes, nt); break;
            case 8: rc = resident_launch<8>(ctx, ra, red, grid, lds_bytes, nt); break;
            case 24: rc = resident_launch<24>(ctx, ra, red, grid, lds_bytes, nt); break;
            case 40: rc = resident_launch<40>(ctx, ra, red, grid, lds_bytes, nt); break;
            default:
                rc = hybrid ? resident_launch<ER_MAX, true>(ctx, ra, red, grid, lds_bytes, nt)
                            : resident_launch<ER_MAX>(ctx, ra, red, grid, lds_bytes, nt);
                break;
        }
    }
    if (rc != LBFGS_HIP_OK) return rc;
    if (gated) {  // the chain that closes this launch's hand-offs across the ranks, behind the kernel that is already running
        const int rc_c = enqueue_gated_chain(ctx, first_exchange, (int)handoffs, owl ? 4 : 2);
        if (rc_c != LBFGS_HIP_OK) return rc_c;
    }
#if LH_RES_TRACE
    res_trace_collect(ctx, trace_first_tag, (int)handoffs, grid);
#endif
    ctx->resident_launches += 1;
    ctx->resident_elements = hybrid ? 2ull * per_round * (uint64_t)(er + el) : n;
    *new_end = e1;
    return 1;
}
}  // namespace


// One two-loop recursion: the persistent on-chip kernel where the shard is eligible (resident.h), else a kernel per step.
// (Replaying the kernel-per-step chain as a hipGraph was built in round 2 and measured 0-2 % SLOWER than eager launches at
// every size -- the chain is bound by the GPU-side cost of a dependent kernel boundary, not by the host's launches -- and
// was removed in round 4: profiles/r02_graph_vs_eager.log, EXPERIMENTS.md.)
static int two_loop_impl_inner(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                               int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                               bool owl, uint64_t owl_start, uint64_t owl_end);
static int two_loop_impl(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                         int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                         bool owl, uint64_t owl_start, uint64_t owl_end) {
    if (!h) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    ctx->two_loop_calls += 1;
    ctx->xchg_class = 1;  // (the exchanges enqueued from here on are charged to the two-loop: lbfgs_hip_ctx_comm_info)
    const int rc = two_loop_impl_inner(h, d, g, k, end, gamma_num_slot, gamma_den_slot, dnorm_slot, first_dot_slot, new_end, owl,
                                       owl_start, owl_end);
    ctx->xchg_class = 0;
    return rc;
}
static int two_loop_impl_inner(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                               int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                               bool owl, uint64_t owl_start, uint64_t owl_end) {
    if (first_dot_slot >= 0 && !slot_ok(first_dot_slot, 1)) return LBFGS_HIP_ERR_ARG;
    if (!h || !d || !g || d->ctx != h->ctx || g->ctx != h->ctx || end < 0 || end >= h->m || !new_end)
        return LBFGS_HIP_ERR_ARG;
    if (!slot_ok(gamma_num_slot, 1) || !slot_ok(gamma_den_slot, 1) || !slot_ok(dnorm_slot, 2)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    {  // small enough to keep the running vector on the chip?  then the whole recursion is one kernel
        const int rr = two_loop_resident(h, d, g, k, end, gamma_num_slot, gamma_den_slot, dnorm_slot, first_dot_slot, new_end, owl,
                                         owl_start, owl_end);
        if (rr > 0 && ctx->comm_kind == LBFGS_HIP_COMM_NONE) {  // (see lbfgs_hip_ctx::last_res)
            lbfgs_hip_ctx::LastResident& lr = ctx->last_res;
            lr.valid = true; lr.h = h; lr.d = d; lr.g = g; lr.k = k; lr.end = end;
            lr.gnum = gamma_num_slot; lr.gden = gamma_den_slot; lr.dn = dnorm_slot; lr.first = first_dot_slot;
            lr.owl = owl; lr.owl_start = owl_start; lr.owl_end = owl_end;
        }
        if (rr != 0) return rr < 0 ? rr : LBFGS_HIP_OK;
    }
    return two_loop_eager(h, d, g, k, end, gamma_num_slot, gamma_den_slot, dnorm_slot, first_dot_slot, new_end, owl, owl_start, owl_end);
}

int two_loop_eager(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                   int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                   bool owl, uint64_t owl_start, uint64_t owl_end) {
    lbfgs_hip_ctx* ctx = h->ctx;
    ProfScope whole(ctx, LBFGS_HIP_K_TWOLOOP_ALL);
    const int m = h->m;
    const int e1 = (end + 1) % m;                                   // lbfgs.rs:577
    const int bound = (int)((uint64_t)m < k ? (uint64_t)m : k);     // lbfgs.rs:579
    *new_end = e1;
    const double* gnum = ctx->board + gamma_num_slot;
    const double* gden = ctx->board + gamma_den_slot;
    double* dn = ctx->board + dnorm_slot;
    double* dots = ctx->board + LBFGS_HIP_BOARD_SLOTS;  // 2 private ping-pong slots past the public board
    if (bound == 0) {
        // no corrections yet: d = (-g)*gamma (core.rs:95-101, lbfgs.rs:591), ||d||^2 and g.d
        OpScaleDevNorm op{};
        op.in[0] = g->p; op.out[0] = d->p; op.gn = gnum; op.gd = gden;
        double* outs[2] = {dn, dn + 1};
        int rc0 = launch(ctx, LBFGS_HIP_K_TWOLOOP_EDGE, op, outs);
        if (rc0 != LBFGS_HIP_OK || !owl) return rc0;
        return lbfgs_hip_constrain_direction(d, g, owl_start, owl_end, dnorm_slot + 2);
    }
    // slot visited at first-loop step i (i = 0..bound-1): j_i = (e1 - 1 - i) mod m   (lbfgs.rs:583)
    auto jat = [&](int i) { return ((e1 - 1 - i) % m + m) % m; };
    int pp = 0;
    int rc;
    // Inner dot products: with one rank and the tagged hand-off nobody reduces them inside the producing kernel -- its
    // workgroups leave their partial sums and exit, and the next step adds them up in its prologue (same order, same
    // bits), where that work overlaps the first loads.  Saves the reducer's poll + sum + store on 2*bound - 1 dependent
    // kernels.  With several ranks the totals have to exist before they can be exchanged, so the reducer stays.
    const bool defer = ctx->comm_kind == LBFGS_HIP_COMM_NONE && !ctx->handoff_ticket && ctx->defer_inner_sums;
    auto slot_of = [&](int par) {
        DotRef r;
        if (defer) r.parts = ctx->dot_parts + (size_t)par * MAX_GRID;
        else r.scalar = dots + par;
        return r;
    };
    DotRef cur;  // where the pending numerator lives
    if (first_dot_slot >= 0) {
        // alpha_0 numerator already on the board (the history-update kernel summed s_new.(-g)): no pass at all
        cur.scalar = ctx->board + first_dot_slot;
    } else {   // alpha_0 numerator: s_{j0} . (-g)                                         2r
        OpTwoLoopFirst op{};
        op.in[0] = g->p; op.in[1] = h->s[jat(0)]->p;
        cur = slot_of(pp);
        double* outs[1] = {defer ? cur.parts : cur.scalar};
        rc = launch(ctx, LBFGS_HIP_K_TWOLOOP_EDGE, op, outs, nullptr, 0, defer, defer ? &cur.nparts : nullptr);
        if (rc != LBFGS_HIP_OK) return rc;
    }
    // first loop, steps 1..bound-1: q -= alpha_{i-1} y_{j_{i-1}} ; next numerator s_{j_i} . q     3r 1w
    for (int i = 1; i < bound; ++i) {
        const int jp = jat(i - 1), jn = jat(i);
        const double* src = (i == 1) ? g->p : d->p;
        DotRef nxt = slot_of(pp ^ 1);
        if (i == 1)
            rc = two_loop_step<true, false, 0>(h, src, h->y[jp]->p, h->s[jn]->p, d->p, cur, jp, 0, gnum, gden, &nxt,
                                               LBFGS_HIP_K_TWOLOOP_STEP);
        else
            rc = two_loop_step<false, false, 0>(h, src, h->y[jp]->p, h->s[jn]->p, d->p, cur, jp, 0, gnum, gden, &nxt,
                                                LBFGS_HIP_K_TWOLOOP_STEP);
        if (rc != LBFGS_HIP_OK) return rc;
        pp ^= 1;
        cur = nxt;
    }
    {   // transition: q = gamma*(q - alpha_last y_last) ; beta numerator y_last . q      2r 1w
        const int jl = jat(bound - 1);
        DotRef nxt = slot_of(pp ^ 1);
        if (bound == 1)
            rc = two_loop_step<true, true, 1>(h, g->p, h->y[jl]->p, nullptr, d->p, cur, jl, 0, gnum, gden, &nxt,
                                              LBFGS_HIP_K_TWOLOOP_EDGE);
        else
            rc = two_loop_step<false, true, 1>(h, d->p, h->y[jl]->p, nullptr, d->p, cur, jl, 0, gnum, gden, &nxt,
                                               LBFGS_HIP_K_TWOLOOP_EDGE);
        if (rc != LBFGS_HIP_OK) return rc;
        pp ^= 1;
        cur = nxt;
    }
    // second loop (lbfgs.rs:594-601), slots in the reverse order: j = jat(bound-1) ... jat(0)
    for (int i = bound - 1; i >= 1; --i) {
        const int j = jat(i), jn = jat(i - 1);  // q += (alpha_j - beta_j) s_j ; next numerator y_{jn} . q   3r 1w
        DotRef nxt = slot_of(pp ^ 1);
        rc = two_loop_step<false, false, 0>(h, d->p, h->s[j]->p, h->y[jn]->p, d->p, cur, j, 1, gnum, gden, &nxt,
                                            LBFGS_HIP_K_TWOLOOP_STEP);
        if (rc != LBFGS_HIP_OK) return rc;
        pp ^= 1;
        cur = nxt;
    }
    // last step: q += (alpha_{j0} - beta_{j0}) s_{j0} ; ||d||^2 (lbfgs.rs:543) and g.d (core.rs:78-92)   3r 1w
    DotRef fin;
    fin.scalar = dn;
    if (owl)  // ... plus the orthant projection of d and the post-projection ||d||^2, pg.d (orthantwise.rs:140-161)
        return two_loop_step<false, false, 3>(h, d->p, h->s[jat(0)]->p, g->p, d->p, cur, jat(0), 1, gnum, gden, &fin,
                                              LBFGS_HIP_K_TWOLOOP_EDGE, owl_start, owl_end);
    return two_loop_step<false, false, 2>(h, d->p, h->s[jat(0)]->p, g->p, d->p, cur, jat(0), 1, gnum, gden, &fin,
                                          LBFGS_HIP_K_TWOLOOP_EDGE);
}

extern "C" {

// Unfused reference sequence (lbfgs.rs:582-601 as written: dot, axpy, ..., scale, dot, axpy, ...)

int lbfgs_hip_two_loop_unfused(lbfgs_hip_history* h, lbfgs_hip_vec* d, uint64_t k, int end, int gamma_num_slot,
                               int gamma_den_slot, int* new_end) {
    if (!h || !d || d->ctx != h->ctx || end < 0 || end >= h->m || !new_end) return LBFGS_HIP_ERR_ARG;
    if (!slot_ok(gamma_num_slot, 1) || !slot_ok(gamma_den_slot, 1)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    const int m = h->m;
    const int e1 = (end + 1) % m;
    const int bound = (int)((uint64_t)m < k ? (uint64_t)m : k);
    *new_end = e1;
    double* dot = ctx->board + LBFGS_HIP_BOARD_SLOTS;
    int j = e1, rc;
    for (int it = 0; it < bound; ++it) {
        j = (j + m - 1) % m;
        OpDot dt{};
        dt.in[0] = h->s[j]->p; dt.in[1] = d->p;
        double* outs[1] = {dot};
        if ((rc = launch(ctx, LBFGS_HIP_K_BLAS1, dt, outs)) != LBFGS_HIP_OK) return rc;
        OpAxpyAlpha ax{};
        ax.in[0] = d->p; ax.in[1] = h->y[j]->p; ax.out[0] = d->p;
        ax.dot_in = dot; ax.ys_j = h->ys + j; ax.alpha_j = h->alpha + j; ax.mode_b = 0;
        if ((rc = launch(ctx, LBFGS_HIP_K_BLAS1, ax, nullptr)) != LBFGS_HIP_OK) return rc;
    }
    OpScaleDev sc{};
    sc.in[0] = d->p; sc.out[0] = d->p; sc.gn = ctx->board + gamma_num_slot; sc.gd = ctx->board + gamma_den_slot;
    if ((rc = launch(ctx, LBFGS_HIP_K_BLAS1, sc, nullptr)) != LBFGS_HIP_OK) return rc;
    for (int it = 0; it < bound; ++it) {
        OpDot dt{};
        dt.in[0] = h->y[j]->p; dt.in[1] = d->p;
        double* outs[1] = {dot};
        if ((rc = launch(ctx, LBFGS_HIP_K_BLAS1, dt, outs)) != LBFGS_HIP_OK) return rc;
        OpAxpyAlpha ax{};
        ax.in[0] = d->p; ax.in[1] = h->s[j]->p; ax.out[0] = d->p;
        ax.dot_in = dot; ax.ys_j = h->ys + j; ax.alpha_j = h->alpha + j; ax.mode_b = 1;
        if ((rc = launch(ctx, LBFGS_HIP_K_BLAS1, ax, nullptr)) != LBFGS_HIP_OK) return rc;
        j = (j + 1) % m;
    }
    return LBFGS_HIP_OK;
}

// vector-free (Gram) variant of lbfgs_hip_two_loop: same arguments, same results up to rounding
int lbfgs_hip_two_loop_gram(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                            int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int* new_end) {
    if (!h || !d || !g || d->ctx != h->ctx || g->ctx != h->ctx || end < 0 || end >= h->m || !new_end)
        return LBFGS_HIP_ERR_ARG;
    if (!slot_ok(gamma_num_slot, 1) || !slot_ok(gamma_den_slot, 1) || !slot_ok(dnorm_slot, 4)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    const int m = h->m;
    if (m > GRAM_MAX_M) return fail(ctx, LBFGS_HIP_ERR_ARG, "vector-free two-loop supports m <= %d", GRAM_MAX_M);
    if (!h->gram) {
        const size_t nb = 2 * (size_t)m + 1, words = nb * nb + 3 * nb + nb + 2;
        HIP_TRY(ctx, hipMalloc(&h->gram, words * sizeof(double)));
        HIP_TRY(ctx, hipMemsetAsync(h->gram, 0, words * sizeof(double), ctx->stream));
        h->gram_rows = h->gram + nb * nb;
        h->gram_delta = h->gram_rows + 3 * nb;
        h->gram_pred = h->gram_delta + nb;
    }
    ProfScope whole(ctx, LBFGS_HIP_K_TWOLOOP_ALL);
    const int e1 = (end + 1) % m;
    const int bound = (int)((uint64_t)m < k ? (uint64_t)m : k);
    *new_end = e1;
    const double* gnum = ctx->board + gamma_num_slot;
    const double* gden = ctx->board + gamma_den_slot;
    double* dn = ctx->board + dnorm_slot;
    ctx->two_loop_calls += 1;
    ctx->xchg_class = 1;
    int rc = LBFGS_HIP_ERR_ARG;
    switch (m) {
#define LH_CASE(MM) case MM: rc = two_loop_gram_impl<MM>(h, d, g, bound, end, e1, gnum, gden, dn); break;
        LH_CASE(1) LH_CASE(2) LH_CASE(3) LH_CASE(4) LH_CASE(5) LH_CASE(6)
        LH_CASE(7) LH_CASE(8) LH_CASE(9) LH_CASE(10)
#undef LH_CASE
        default: break;
    }
    ctx->xchg_class = 0;
    return rc;
}

// ==================================================================================== OWL-QN
int lbfgs_hip_owlqn_post_eval(const lbfgs_hip_vec* x, const lbfgs_hip_vec* g, lbfgs_hip_vec* pg, double c,
                              uint64_t start, uint64_t end, int out_slot) {
    if (!same_ctx(x, g) || !same_ctx(x, pg) || !slot_ok(out_slot, 3)) return LBFGS_HIP_ERR_ARG;
    OpOwlPost op{};
    op.in[0] = x->p; op.in[1] = g->p; op.out[0] = pg->p;
    op.c = c; op.start = start; op.end = end;
    double* b = x->ctx->board + out_slot;
    double* outs[3] = {b, b + 1, b + 2};
    return launch(x->ctx, LBFGS_HIP_K_OWLQN, op, outs);
}

int lbfgs_hip_orthant_select(lbfgs_hip_vec* wp, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* pg) {
    if (!same_ctx(wp, xp) || !same_ctx(wp, pg)) return LBFGS_HIP_ERR_ARG;
    OpOrthantSelect op{};
    op.in[0] = xp->p; op.in[1] = pg->p; op.out[0] = wp->p;
    return launch(wp->ctx, LBFGS_HIP_K_OWLQN, op, nullptr);
}

int lbfgs_hip_constrain_direction(lbfgs_hip_vec* d, const lbfgs_hip_vec* pg, uint64_t start, uint64_t end,
                                  int out_slot) {
    if (!same_ctx(d, pg) || !slot_ok(out_slot, 2)) return LBFGS_HIP_ERR_ARG;
    OpConstrainDir op{};
    op.in[0] = d->p; op.in[1] = pg->p; op.out[0] = d->p;
    op.start = start; op.end = end;
    double* outs[2] = {d->ctx->board + out_slot, d->ctx->board + out_slot + 1};
    return launch(d->ctx, LBFGS_HIP_K_OWLQN, op, outs);
}

// ==================================================================================== objectives
int lbfgs_hip_objective_eval(const lbfgs_hip_objective* obj, const lbfgs_hip_vec* x, lbfgs_hip_vec* g, int out_slot) {
    if (!obj || !same_ctx(x, g) || !slot_ok(out_slot, 1)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = x->ctx;
    double* outs[1] = {ctx->board + out_slot};
    switch (obj->kind) {
        case LBFGS_HIP_OBJ_QUADRATIC: {
            OpObjEval<ObjQuadratic> op{};
            op.in[0] = x->p; op.out[0] = g->p; op.obj = {obj->seed_a, obj->seed_b};
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        case LBFGS_HIP_OBJ_LOGISTIC: {
            OpObjEval<ObjLogistic> op{};
            op.in[0] = x->p; op.out[0] = g->p; op.obj = {obj->seed_a, obj->seed_b};
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        case LBFGS_HIP_OBJ_ROSENBROCK: {
            int rc = rosen_shape_ok(ctx);
            if (rc != LBFGS_HIP_OK) return rc;
            OpRosenEval op{};
            op.in[0] = x->p; op.out[0] = g->p;
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        case LBFGS_HIP_OBJ_LJ_ALLPAIRS:
        case LBFGS_HIP_OBJ_LJ_NEIGHBORS:
        case LBFGS_HIP_OBJ_LJ_CELLS:
            return lj_eval(ctx, obj, x, g, outs[0]);
        default:
            return fail(ctx, LBFGS_HIP_ERR_ARG, "unknown objective kind %d", obj->kind);
    }
}

int lbfgs_hip_objective_line_eval(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                  const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* g, int out_slot) {
    if (!obj || !same_ctx(x, xp) || !same_ctx(x, d) || !same_ctx(x, g) || !slot_ok(out_slot, 2)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = x->ctx;
    double* outs[2] = {ctx->board + out_slot, ctx->board + out_slot + 1};
    switch (obj->kind) {
        case LBFGS_HIP_OBJ_QUADRATIC: {
            OpObjLineEval<ObjQuadratic> op{};
            op.in[0] = xp->p; op.in[1] = d->p; op.out[0] = x->p; op.out[1] = g->p;
            op.step = step; op.obj = {obj->seed_a, obj->seed_b};
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        case LBFGS_HIP_OBJ_LOGISTIC: {
            OpObjLineEval<ObjLogistic> op{};
            op.in[0] = xp->p; op.in[1] = d->p; op.out[0] = x->p; op.out[1] = g->p;
            op.step = step; op.obj = {obj->seed_a, obj->seed_b};
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        case LBFGS_HIP_OBJ_LJ_CELLS:
            if (ctx->lj_fused_trial) {  // the trial point is formed by the list check's pass, g.d summed where g is formed (lj.hip)
                ctx->last_res.valid = false;
                const LjTrial tr{xp->p, d->p, step, outs[1]};
                return lj_eval(ctx, obj, x, g, outs[0], &tr);
            }
            [[fallthrough]];
        case LBFGS_HIP_OBJ_LJ_ALLPAIRS:
        case LBFGS_HIP_OBJ_LJ_NEIGHBORS: {  // gather pattern: line step, evaluate, g.d as three launches
            int rc = lbfgs_hip_line_step(x, xp, d, step, nullptr, 0, 0);
            if (rc != LBFGS_HIP_OK) return rc;
            if ((rc = lj_eval(ctx, obj, x, g, outs[0])) != LBFGS_HIP_OK) return rc;
            return lbfgs_hip_vecdot(g, d, out_slot + 1);
        }
        case LBFGS_HIP_OBJ_ROSENBROCK: {
            int rc = rosen_shape_ok(ctx);
            if (rc != LBFGS_HIP_OK) return rc;
            OpRosenLineEval op{};
            op.in[0] = xp->p; op.in[1] = d->p; op.out[0] = x->p; op.out[1] = g->p;
            op.step = step;
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        default:
            return fail(ctx, LBFGS_HIP_ERR_ARG, "unknown objective kind %d", obj->kind);
    }
}

int lbfgs_hip_objective_is_elementwise(const lbfgs_hip_objective* obj) {
    return obj && (obj->kind == LBFGS_HIP_OBJ_QUADRATIC || obj->kind == LBFGS_HIP_OBJ_LOGISTIC);
}

int lbfgs_hip_objective_line_probe(const lbfgs_hip_objective* obj, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d,
                                   double step, int out_slot) {
    if (!obj || !same_ctx(xp, d) || !slot_ok(out_slot, 2)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = xp->ctx;
    double* outs[2] = {ctx->board + out_slot, ctx->board + out_slot + 1};
    switch (obj->kind) {
        case LBFGS_HIP_OBJ_QUADRATIC: {
            OpObjLineProbe<ObjQuadratic> op{};
            op.in[0] = xp->p; op.in[1] = d->p; op.step = step; op.obj = {obj->seed_a, obj->seed_b};
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        case LBFGS_HIP_OBJ_LOGISTIC: {
            OpObjLineProbe<ObjLogistic> op{};
            op.in[0] = xp->p; op.in[1] = d->p; op.step = step; op.obj = {obj->seed_a, obj->seed_b};
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        default:
            return fail(ctx, LBFGS_HIP_ERR_ARG, "objective kind %d is not evaluated element by element", obj->kind);
    }
}

}  // extern "C"

template <class Obj, bool DAMP>
static int update_from_step(lbfgs_hip_history* h, int slot, const lbfgs_hip_objective* obj, lbfgs_hip_vec* x,
                            const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double t, lbfgs_hip_vec* g,
                            const lbfgs_hip_vec* gp, double step, double** outs, bool single) {
    OpHistUpdateFromStep<Obj, DAMP> op{};
    op.in[0] = xp->p; op.in[1] = d->p; op.in[2] = gp->p;
    op.out[0] = x->p; op.out[1] = g->p; op.out[2] = h->s[slot]->p; op.out[3] = h->y[slot]->p;
    op.t = t;
    op.neg_step = DAMP ? -step : 0.0;
    op.obj = {obj->seed_a, obj->seed_b};
    return launch(h->ctx, LBFGS_HIP_K_UPDATE, op, outs, single ? h->ys + slot : nullptr, 1);
}

extern "C" {

int lbfgs_hip_history_update_from_step(lbfgs_hip_history* h, int slot, const lbfgs_hip_objective* obj,
                                       lbfgs_hip_vec* x, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double t,
                                       lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, double step, int damping,
                                       int out_slot) {
    if (!h || slot < 0 || slot >= h->m || !obj || !x || !xp || !d || !g || !gp) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    if (x->ctx != ctx || xp->ctx != ctx || d->ctx != ctx || g->ctx != ctx || gp->ctx != ctx) return LBFGS_HIP_ERR_ARG;
    if (!slot_ok(out_slot, 7)) return LBFGS_HIP_ERR_ARG;
    if (x == xp || g == gp) return fail(ctx, LBFGS_HIP_ERR_ARG, "update_from_step: x/g must not alias xp/gp");
    double* b = ctx->board + out_slot;
    double* outs[7] = {b, b + 1, b + 2, b + 3, b + 4, b + 5, b + 6};
    const bool single = ctx->comm_kind == LBFGS_HIP_COMM_NONE || ctx->comm_kind == LBFGS_HIP_COMM_P2P;
    int rc;
    switch (obj->kind) {
        case LBFGS_HIP_OBJ_QUADRATIC:
            rc = damping ? update_from_step<ObjQuadratic, true>(h, slot, obj, x, xp, d, t, g, gp, step, outs, single)
                         : update_from_step<ObjQuadratic, false>(h, slot, obj, x, xp, d, t, g, gp, step, outs, single);
            break;
        case LBFGS_HIP_OBJ_LOGISTIC:
            rc = damping ? update_from_step<ObjLogistic, true>(h, slot, obj, x, xp, d, t, g, gp, step, outs, single)
                         : update_from_step<ObjLogistic, false>(h, slot, obj, x, xp, d, t, g, gp, step, outs, single);
            break;
        default:
            return fail(ctx, LBFGS_HIP_ERR_ARG, "objective kind %d is not evaluated element by element", obj->kind);
    }
    if (rc != LBFGS_HIP_OK) return rc;
    if (!single)  // lbfgs.rs:656 self.ys = the GLOBAL y.s (see lbfgs_hip_history_update)
        HIP_TRY(ctx, hipMemcpyAsync(h->ys + slot, b + 1, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return LBFGS_HIP_OK;
}

}  // extern "C"  (the template below needs C++ linkage)

namespace {
// one OWL-QN trial; FIRST: wp is an OUTPUT formed from xp and the previous pg (ops.h OpObjOwlLineEval<Obj, true>);
// UPD: the trial also fills history slot `slot` (s, y, ys) and leaves ||s||^2, y.s, y.y at board[upd_slot ..]
template <bool FIRST, bool UPD>
int owlqn_trial(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double step,
                lbfgs_hip_vec* wp, lbfgs_hip_vec* g, lbfgs_hip_vec* pg, double c, uint64_t start, uint64_t end, int out_slot,
                lbfgs_hip_history* h = nullptr, int slot = 0, const lbfgs_hip_vec* gp = nullptr, int upd_slot = 0) {
    if (!obj || !same_ctx(x, xp) || !same_ctx(x, d) || !same_ctx(x, wp) || !same_ctx(x, g) || !same_ctx(x, pg) ||
        !slot_ok(out_slot, 5))
        return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = x->ctx;
    if (UPD && (!h || h->ctx != ctx || slot < 0 || slot >= h->m || !same_ctx(x, gp) || !slot_ok(upd_slot, 3))) return LBFGS_HIP_ERR_ARG;
    double* b = ctx->board + out_slot;
    double* u = ctx->board + upd_slot;
    double* outs[8] = {b, b + 1, b + 2, b + 3, b + 4, u, u + 1, u + 2};
    // (UPD) lbfgs.rs:656 self.ys = ys: stored by the kernel itself when its totals are final, else copied after the all-reduce
    const bool single = ctx->comm_kind == LBFGS_HIP_COMM_NONE || ctx->comm_kind == LBFGS_HIP_COMM_P2P;
    auto fill = [&](auto& op) {
        op.in[0] = xp->p; op.in[1] = d->p; op.in[2] = FIRST ? pg->p : wp->p;
        op.out[0] = x->p; op.out[1] = g->p; op.out[2] = pg->p;
        if (FIRST) op.out[3] = wp->p;
        if (UPD) {
            op.in[3] = gp->p;
            op.out[op.O_S] = h->s[slot]->p;
            op.out[op.O_S + 1] = h->y[slot]->p;
        }
        op.step = step; op.c = c; op.start = start; op.end = end; op.obj = {obj->seed_a, obj->seed_b};
    };
    int rc;
    switch (obj->kind) {
        case LBFGS_HIP_OBJ_QUADRATIC: {
            OpObjOwlLineEval<ObjQuadratic, FIRST, UPD> op{};
            fill(op);
            rc = launch(ctx, LBFGS_HIP_K_EVAL, op, outs, (UPD && single) ? h->ys + slot : nullptr, 6);
            break;
        }
        case LBFGS_HIP_OBJ_LOGISTIC: {
            OpObjOwlLineEval<ObjLogistic, FIRST, UPD> op{};
            fill(op);
            rc = launch(ctx, LBFGS_HIP_K_EVAL, op, outs, (UPD && single) ? h->ys + slot : nullptr, 6);
            break;
        }
        default: {  // objectives without a fused form: the separate passes
            rc = LBFGS_HIP_OK;
            if (FIRST && (rc = lbfgs_hip_orthant_select(wp, xp, pg)) != LBFGS_HIP_OK) return rc;
            if ((rc = lbfgs_hip_line_step(x, xp, d, step, wp, start, end)) != LBFGS_HIP_OK) return rc;
            if ((rc = lbfgs_hip_objective_eval(obj, x, g, out_slot)) != LBFGS_HIP_OK) return rc;
            if ((rc = lbfgs_hip_owlqn_post_eval(x, g, pg, c, start, end, out_slot + 2)) != LBFGS_HIP_OK) return rc;
            if ((rc = lbfgs_hip_vecdot(g, d, out_slot + 1)) != LBFGS_HIP_OK) return rc;
            if (UPD) {  // (the update's own pass; its sums 3..6 land in a scratch copy of the layout: only the first three are asked for)
                if (!slot_ok(upd_slot, 7)) return LBFGS_HIP_ERR_ARG;
                return lbfgs_hip_history_update(h, slot, x, xp, g, gp, step, 0, upd_slot);
            }
            return rc;
        }
    }
    if (rc != LBFGS_HIP_OK) return rc;
    if (UPD && !single) HIP_TRY(ctx, hipMemcpyAsync(h->ys + slot, u + 1, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return LBFGS_HIP_OK;
}
}  // namespace

extern "C" {

int lbfgs_hip_objective_owlqn_line_eval(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                        const lbfgs_hip_vec* d, double step, const lbfgs_hip_vec* wp, lbfgs_hip_vec* g,
                                        lbfgs_hip_vec* pg, double c, uint64_t start, uint64_t end, int out_slot) {
    return owlqn_trial<false, false>(obj, x, xp, d, step, const_cast<lbfgs_hip_vec*>(wp), g, pg, c, start, end, out_slot);
}

int lbfgs_hip_objective_owlqn_first_trial(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                          const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* wp, lbfgs_hip_vec* g,
                                          lbfgs_hip_vec* pg, double c, uint64_t start, uint64_t end, int out_slot) {
    return owlqn_trial<true, false>(obj, x, xp, d, step, wp, g, pg, c, start, end, out_slot);
}

int lbfgs_hip_objective_owlqn_trial_update(const lbfgs_hip_objective* obj, lbfgs_hip_history* h, int slot, lbfgs_hip_vec* x,
                                           const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* wp,
                                           int first, lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, lbfgs_hip_vec* pg, double c,
                                           uint64_t start, uint64_t end, int out_slot, int upd_slot) {
    return first ? owlqn_trial<true, true>(obj, x, xp, d, step, wp, g, pg, c, start, end, out_slot, h, slot, gp, upd_slot)
                 : owlqn_trial<false, true>(obj, x, xp, d, step, wp, g, pg, c, start, end, out_slot, h, slot, gp, upd_slot);
}

}  // extern "C"
