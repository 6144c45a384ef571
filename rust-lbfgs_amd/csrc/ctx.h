// rust-lbfgs_amd/csrc/ctx.h -- what the three translation units of liblbfgs_hip.so share on the HOST side: the objects behind
// the opaque handles of include/lbfgs_hip.h (context, vector, history), the error / launch plumbing, and the few internal
// functions one unit needs from another.  Not installed, not part of the ABI.
//     context.hip    device, stream, board, communicators (RCCL, P2P mailboxes, callback), pools, vectors, scalars, timing
//     lbfgs_hip.hip  the launchers: primitives, fused operators, history, the two-loop in all its forms, objectives
//     lj.hip         the Lennard-Jones evaluators' bookkeeping (cell list, Verlet list, rebuilds)
#pragma once
#include "../../include/lbfgs_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "stream.h"

using namespace lh;

#define LH_INTERNAL __attribute__((visibility("hidden")))  // shared between the units, not exported from the library

typedef struct { char internal[128]; } nccl_unique_id_t;
typedef void* nccl_comm_t;

// ------------------------------------------------------------------------------------ objects
struct ProfPair {
    hipEvent_t a, b;
};
struct ProfClass {
    std::vector<ProfPair> pending;
    uint64_t launches = 0;
    double ms = 0.0;
};

// LJ_CELLS (lj.h): device buffers and bookkeeping of the rebuildable neighbour list of ONE context
struct LjCells {
    uint32_t natoms = 0, max_nbr = 0;
    double cutoff = 0.0, skin = 0.0;
    size_t ncap = 0;                 // cells the count / start / cursor arrays can hold
    int32_t* nbr_rows = nullptr;     // the list as the build kernel writes it: [natoms][max_nbr]
    int32_t* slot_of = nullptr;      // an atom's place within its cell (from the count kernel's atomicAdd)
    int32_t *nbr = nullptr, *cnt = nullptr, *cell_of = nullptr, *sorted = nullptr, *counts = nullptr, *starts = nullptr,
            *cursor = nullptr;
    double* xref = nullptr;          // positions at build time
    double* xs = nullptr;            // positions (+ atom index) in cell order, 32 bytes per atom (build time only)
    int32_t* tile_sums = nullptr;    // scan scratch, 1024 entries
    double* bbox_part = nullptr;     // [LJ_BBOX_GRID][6]
    unsigned int* overflow = nullptr;
    double* host = nullptr;          // host-mapped: [0] = atoms outside their skin/2 sphere (written by the evaluation kernel)
    double* host_dev = nullptr;
    bool built = false;
    bool transposed = false;         // nbr holds the column-major form of nbr_rows
    uint64_t evals_this_list = 0, last_lifetime = 0;  // successful evaluations over the current / the previous list
    uint64_t rebuilds = 0, evals = 0;
    uint32_t longest = 0;
    void release() {
        void* ptrs[] = {nbr, nbr_rows, slot_of, cnt, cell_of, sorted, counts, starts, cursor, xref, xs, tile_sums, bbox_part, overflow};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (host) (void)hipHostFree(host);
        *this = LjCells();
    }
};
constexpr int LJ_BBOX_GRID = 256;

struct lbfgs_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    lbfgs_hip_shard shard{};
    int comm_kind = LBFGS_HIP_COMM_NONE;
    nccl_comm_t nccl = nullptr;
    int rccl_ranks_seen = 0, rccl_rank_seen = -1;   // ncclCommCount / ncclCommUserRank of `nccl` (checked against the shard)
    int p2p_peers_device = 0, p2p_peers_host = 0;    // peers' mailboxes mapped at context creation, by placement
    unsigned long long two_loop_calls = 0;           // two-loop recursions enqueued (any form) ...
    unsigned long long two_loop_exchanges = 0;       // ... and the cross-rank exchanges enqueued inside them (world > 1)
    unsigned long long allreduce_calls = 0;          // stand-alone all-reduces enqueued (RCCL / callback / P2P kernel)
    int xchg_class = 0;                              // 1 while a two-loop recursion is being enqueued (RedCtl::xchg_class)
    lbfgs_hip_allreduce_cb cb = nullptr;
    void* cb_user = nullptr;
    // P2P communicator
    unsigned long long* p2p_mbox[P2P_MAX_WORLD] = {nullptr};  // [rank] -> that rank's mailbox (own entry = local)
    bool p2p_opened[P2P_MAX_WORLD] = {false};
    unsigned long long p2p_count = 0;     // P2P exchanges enqueued so far (shadow of DevCounters::p2p_epoch)
    unsigned int* p2p_err = nullptr;      // device error flag: 1 = a P2P spin timed out, 2 = a workgroup's partials never arrived
    unsigned long long p2p_timeout_ticks = 0;
    unsigned long long handoff_timeout_ticks = 1000000000ULL;  // bound on every cross-workgroup spin (wall_clock64: 100 MHz)
    // ... except in a context's FIRST resident launches: until one of them has been seen to complete, a hand-off waits 50 ms
    // at most (LBFGS_HIP_RESIDENT_FIRST_TIMEOUT_MS), so that a device which cannot hold the grid resident -- partitioned, shared
    // with another resident kernel, CU-masked in a way the probes miss -- costs milliseconds, not the full timeout, before the
    // kernel-per-step path takes over (lbfgs_hip_scalars_read)
    unsigned long long first_timeout_ticks = 5000000ULL;
    bool resident_proven = false;
    double* board = nullptr;         // LBFGS_HIP_BOARD_SLOTS doubles + 2 ping-pong dots
    double* partials = nullptr;      // MAX_RED * MAX_GRID
    unsigned int* ticket = nullptr;
    double* lj_scratch = nullptr;         // all-pairs LJ: per-j-range force slices (allocated on demand)
    size_t lj_scratch_bytes = 0;
    struct LjCells* lj_cells = nullptr;   // LJ_CELLS: the rebuildable neighbour structure (allocated on demand)
    unsigned long long* gran = nullptr;   // tagged partial granules [MAX_RED][MAX_GRID][2] (stream.h)
    bool gran_pooled = false;             // ... in an uncached block of the process-wide pool (returned there, never freed)
    bool p2p_exclusive = false;           // lbfgs_hip_comm.exclusive_device: no other rank shares this GPU
    // GATED exchange (RCCL under the persistent two-loop kernel; stream.h ext_exchange, lbfgs_hip.hip enqueue_gated_chain)
    int rccl_resident = 0;                // LBFGS_HIP_RCCL_RESIDENT=1 OPTS IN to the gated exchange under RCCL (the persistent two-loop kernel
                                          // served by ncclAllReduce on a second stream).  Off by default: it has never run with more than
                                          // one rank (a single GPU hosts a 1-rank communicator only); bench.py's "rccl" leg opts in inside a
                                          // child job with a timeout.  Off: under RCCL the two-loop runs with a kernel per step
    hipStream_t xstream = nullptr;        // the second stream: gate -> ncclAllReduce -> post, one triple per exchange
    hipEvent_t ext_ev = nullptr;          // recorded on the compute stream just before a gated persistent kernel; the chain's first gate
                                          // waits for it, so its timeout does not run while the kernel is still queued behind earlier work
    unsigned long long* ext_block = nullptr;  // one uncached block: flags A, B, abort at words 0..2, the ring of exchange slots at byte 256
    unsigned long long ext_launches = 0;  // gated launches so far (the abort word names the launch that is given up)
    unsigned long long resident_attr_mask = 0;  // which resident kernels have had their dynamic-LDS limit raised on this device
    int gram_combine_resident = 1;        // LBFGS_HIP_GRAM_COMBINE_RESIDENT=0: the vector-free combine as one streaming pass over all columns
    int lj_build_fp32 = 1;                // LBFGS_HIP_LJ_BUILD_FP32=0: the LJ_CELLS list from double-precision candidate tests
    int lj_fused_trial = 1;               // LBFGS_HIP_LJ_FUSED_TRIAL=0: an LJ_CELLS trial as line step, evaluate, g.d in three launches
                                          // (the form the other Lennard-Jones evaluators take)
    size_t resident_nt_bytes = (size_t)16 << 20;  // LBFGS_HIP_RESIDENT_NT_MB
    size_t resident_plain_bytes = (size_t)256 << 20;  // LBFGS_HIP_RESIDENT_PLAIN_MB: hybrid: so much of the HBM part of q keeps the default
                                          // cache policy and stays in the 256 MiB Infinity Cache between steps (resident.h res_hbm_rounds)
    int resident_touch = -1;              // LBFGS_HIP_RESIDENT_TOUCH: rounds a waiting workgroup touches ahead (resident.h TOUCHING; -1 = by shard size)
    int resident_hybrid = 1;              // LBFGS_HIP_RESIDENT_HYBRID=0: shards that do not fit the chip take the kernel-per-step path
    uint64_t resident_elements = 0;       // elements of q the last resident launch kept on the chip (all of them unless hybrid)
    int resident_grid = 0;                // LBFGS_HIP_RESIDENT_GRID: workgroups of the resident kernel (0 = one per CU); tests
    bool resident_on = true;              // LBFGS_HIP_RESIDENT=0: never use the on-chip-resident two-loop kernel (resident.h)
    unsigned long long resident_launches = 0;  // two-loops that ran as the resident kernel (tests / bench read it)
    int resident_ok = -1;                 // -1 = not probed yet, 0 = this device / queue cannot hold the grid resident, 1 = usable
    // The latest two-loop if it ran as the resident kernel and nothing has been enqueued since (one rank only): should its
    // hand-offs time out -- the kernel did not get every CU it asked for -- lbfgs_hip_scalars_read re-runs the recursion
    // with a kernel per step (inputs g, s, y, ys are intact; d and alpha are outputs) and stops using the resident kernel.
    struct LastResident {
        bool valid = false;
        lbfgs_hip_history* h = nullptr;
        lbfgs_hip_vec* d = nullptr;
        const lbfgs_hip_vec* g = nullptr;
        uint64_t k = 0, owl_start = 0, owl_end = 0;
        int end = 0, gnum = 0, gden = 0, dn = 0, first = -1;
        bool owl = false;
    } last_res;
    unsigned long long resident_fallbacks = 0;  // how often that happened
    int resident_fault = 0;               // LBFGS_HIP_RESIDENT_FAULT=k (tests): the k-th resident launch of this context loses its last workgroup;
                                          // -1: this rank reports the gated exchange's self-test as failed (context.hip ext_prepare);
                                          // -2 / -3: this rank fails LOCALLY before the handshake / before the trial (it must still
                                          // take part in every collective of ext_prepare and hand the error back afterwards)
    bool defer_inner_sums = true;         // LBFGS_HIP_DEFER_SUMS=0: the two-loop's inner dots are reduced by their own kernels (A/B)
    double* dot_parts = nullptr;          // 2 x MAX_GRID: workgroup partials of the two-loop's inner dot products (ping-pong)
    DevCounters* dev_ctr = nullptr;       // device-resident sequence numbers (stream.h); the three fields below shadow them
    unsigned long long red_count = 0;     // tagged reducing launches enqueued so far: the next one uses tag (red_count % (2^32-1)) + 1
    bool handoff_ticket = false;          // LBFGS_HIP_HANDOFF=ticket: the arrival-counter form for every kernel (A/B, fallback)
    double* pinned = nullptr;        // host staging, LBFGS_HIP_BOARD_SLOTS doubles
    // host mirror of the board (stream.h MirrorCtl)
    double* mirror = nullptr;              // host-mapped: SLOTS+2 doubles, then the sequence word
    double* mirror_dev = nullptr;          // its device address
    unsigned long long mirror_seq = 0;     // sequence number of the latest mirrored launch
    bool mirror_valid[LBFGS_HIP_BOARD_SLOTS + 2] = {false};  // slot's latest value is (or will be) in the mirror
    int grid_default = 0;
    int cu_count = 0;
    int grid_override = 0;
    int grid_x32_class[LBFGS_HIP_K_CLASSES] = {0};  // LBFGS_HIP_GRID_X32_K<class>=<workgroups per 32 CUs>: experiments (0 = the operator's own)
    int gram_grid = 0;  // workgroups of the Gram rows kernel (0 = same as the others)
    size_t nt_threshold_bytes = (size_t)128 << 20;  // measured crossover: 95 MiB vectors prefer plain, 190 MiB prefer nt
    size_t nt_store_threshold_bytes = (size_t)64 << 20;  // from here up to nt_threshold_bytes: `nt` on the STORES only
    bool prof_on = false;
    ProfClass prof[LBFGS_HIP_K_CLASSES];
    std::vector<ProfPair> prof_pool;
    std::string err;
};

struct lbfgs_hip_vec {
    lbfgs_hip_ctx* ctx;
    double* p;
};

struct lbfgs_hip_history {
    lbfgs_hip_ctx* ctx;
    int m;
    std::vector<lbfgs_hip_vec*> s, y;
    double* ys;     // device, m
    double* alpha;  // device, m
    // vector-free (Gram) two-loop: G = B^T B over the basis [s.., y.., g], refreshed rows, coefficients
    double* gram = nullptr;        // (2m+1)^2
    double* gram_rows = nullptr;   // 3*(2m+1), contiguous (one all-reduce message)
    double* gram_delta = nullptr;  // 2m+1
    double* gram_pred = nullptr;   // 2: ||d||^2 predicted by the coefficient-space arithmetic, its cancellation figure (gram_coef_kernel)
};

// ------------------------------------------------------------------------------------ plumbing (context.hip)
LH_INTERNAL int fail(lbfgs_hip_ctx* ctx, int code, const char* fmt, ...);

#define HIP_TRY(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(ctx, LBFGS_HIP_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

inline bool slot_ok(int first, int count) {
    return first >= 0 && count >= 0 && first + count <= LBFGS_HIP_BOARD_SLOTS;
}

LH_INTERNAL int grid_for(const lbfgs_hip_ctx* ctx, int x32 = 27);
// gated exchange: make sure the second stream and the uncached block exist (-> status); give the running chain up and drain it
LH_INTERNAL int ext_prepare(lbfgs_hip_ctx* ctx);
LH_INTERNAL void ext_abort(lbfgs_hip_ctx* ctx);
LH_INTERNAL int rccl_allreduce_on(lbfgs_hip_ctx* ctx, double* buf, int count, hipStream_t stream);
// (lbfgs_hip.hip) one exchange of the gated form under a chip-wide kernel, on the context's own communicator; collective
LH_INTERNAL int ext_selftest_resident(lbfgs_hip_ctx* ctx, bool* ok, bool* enqueued);
LH_INTERNAL bool resident_shard_eligible(lbfgs_hip_ctx* ctx);  // this rank's shard and device can take the persistent kernel under the gated exchange

// ---- profiling: one event pair per launch of a timed class --------------------------------
struct ProfScope {
    lbfgs_hip_ctx* ctx;
    int k;
    ProfPair pr{};
    bool active = false;
    ProfScope(lbfgs_hip_ctx* c, int kclass) : ctx(c), k(kclass) {
        if (!ctx->prof_on) return;
        if (!ctx->prof_pool.empty()) {
            pr = ctx->prof_pool.back();
            ctx->prof_pool.pop_back();
        } else {
            if (hipEventCreate(&pr.a) != hipSuccess || hipEventCreate(&pr.b) != hipSuccess) return;
        }
        active = true;
        (void)hipEventRecord(pr.a, ctx->stream);
    }
    ~ProfScope() {
        if (!active) return;
        (void)hipEventRecord(pr.b, ctx->stream);
        ctx->prof[k].pending.push_back(pr);
    }
};

// the cross-workgroup hand-off part of a RedCtl for the NEXT reducing launch (`nred` sums)
LH_INTERNAL int fill_handoff(lbfgs_hip_ctx* ctx, RedCtl& red, int nred);
// the P2P control block of the NEXT reduction (epochs advance identically on every rank)
LH_INTERNAL P2PCtl next_p2p(lbfgs_hip_ctx* ctx);
// closes a reduction across ranks: `ptrs` are device addresses (board or history scalars) just written by the last workgroup
LH_INTERNAL int allreduce(lbfgs_hip_ctx* ctx, double* const* ptrs, int count);
// the reduction-control block of the NEXT launch with NRED sums (hand-off, P2P exchange, host mirror)
LH_INTERNAL int prep_red(lbfgs_hip_ctx* ctx, RedCtl& red, int nred, double* const* red_out, double* dup_ptr, int dup_k,
                         bool* in_kernel_exchange);
inline bool same_ctx(const lbfgs_hip_vec* a, const lbfgs_hip_vec* b) { return a && b && a->ctx == b->ctx; }

// waits for the stream and looks at the device error word; recovers from a timed-out resident two-loop (context.hip)
LH_INTERNAL int settle(lbfgs_hip_ctx* ctx);

// ------------------------------------------------------------------------------------ across the units
// (lbfgs_hip.hip) the recursion with a kernel per step: what context.hip re-runs after a resident launch has timed out
LH_INTERNAL int two_loop_eager(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                               int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                               bool owl, uint64_t owl_start, uint64_t owl_end);
// (lj.hip) Lennard-Jones objectives: x holds 3*natoms coordinates of ONE rank; f -> *out
// a line-search trial of LJ_CELLS: x = xp + t*d formed by the list check's pass, g.d summed by the evaluation kernel (lj.hip)
struct LjTrial {
    const double* xp;
    const double* d;
    double t;
    double* dg_out;
};
LH_INTERNAL int lj_eval(lbfgs_hip_ctx* ctx, const lbfgs_hip_objective* obj, const lbfgs_hip_vec* x, lbfgs_hip_vec* g, double* out,
                        const LjTrial* step = nullptr);
#if LH_RES_TRACE
LH_INTERNAL void res_trace_print();  // (lbfgs_hip.hip; traced A/B builds only)
#endif
