/*
 * include/lbfgs_hip.h -- C-ABI of the MI355X-native L-BFGS / OWL-QN vector backend.
 *
 * This is the drop-in boundary for the hot path of ybyygu/rust-lbfgs (crate
 * `liblbfgs` v0.2.0).  The reference has NO FFI: its seam
 * is `trait LbfgsMath` (src/math.rs:4-29) plus raw loops in src/core.rs,
 * src/lbfgs.rs and src/orthantwise.rs.  Every entry point below names the
 * reference function (file:line, relative to the crate root) whose body a Rust
 * maintainer would replace by the call; INTEGRATION.md shows the `extern "C"`
 * block and the shim.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures (a stream is passed as void*);
 *  - every function returns an int status (0 = ok, <0 = error, see below) and
 *    never unwinds; lbfgs_hip_last_error() gives the message;
 *  - all n-vectors are f64 and live in HBM for their whole life; the host sees
 *    them only through upload/download;
 *  - reductions (dot, norms, L1 sum) leave their result in a device-resident
 *    SCALAR BOARD (an array of LBFGS_HIP_BOARD_SLOTS doubles owned by the
 *    context); the caller names the slot(s).  Consumers on the device (axpy with
 *    a device coefficient, the two-loop) read the board directly, so a dependent
 *    chain of reductions needs no host round trip.  lbfgs_hip_scalars_read()
 *    is the only call that synchronises the stream;
 *  - with world > 1 the n-vector is sharded contiguously (one process per GPU);
 *    every reduction is closed by a sum all-reduce of its f64 scalars before it
 *    becomes visible in the board, so board values are GLOBAL on every rank;
 *  - element-wise arithmetic keeps the reference's roundings (mul then add, no
 *    FMA contraction); only the summation ORDER of reductions differs from the
 *    reference's sequential sum (src/math.rs:41).  It is a fixed tree: results
 *    are deterministic for a given (n, world).
 */
#ifndef LBFGS_HIP_H
#define LBFGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5: lbfgs_hip_comm_info grew the DISTRIBUTION of the cross-rank exchanges (exchange_us_max, local_wait_us_max,
 *    exchange_hist): one multi-GPU run is then enough to place a leg on the scaling model's latency axis;
 *    lbfgs_hip_objective_owlqn_trial_update (round 6).
 * 4: lbfgs_evaluator (lbfgs_solver.h) grew device_probe / device_accept: deferred trial points for device closures;
 *    lbfgs_hip_objective_owlqn_first_trial (round 5).
 * 3: lbfgs_hip_ctx_comm_info / lbfgs_hip_comm_info (round 4); lbfgs_hip_sync, lbfgs_hip_vec_download and
 *    lbfgs_hip_history_scalars_read report device errors (and recover from a timed-out resident two-loop) like
 *    lbfgs_hip_scalars_read.
 * 2: lbfgs_hip_objective.skin, lbfgs_hip_comm.exclusive_device, LBFGS_HIP_K_TWOLOOP_RESIDENT (all round 2, unversioned then),
 *    mailbox placements, lbfgs_hip_build_id.  A caller compares lbfgs_hip_abi_version() with the constant it was built
 *    against before anything else (rust-lbfgs_amd/_ffi.py, integration/rust-shim, tests/support/c_caller.c do). */
#define LBFGS_HIP_ABI_VERSION 5
#define LBFGS_HIP_BOARD_SLOTS 256

/* status codes */
enum {
    LBFGS_HIP_OK = 0,
    LBFGS_HIP_ERR_ARG = -101,      /* bad argument (null handle, size mismatch, slot out of range) */
    LBFGS_HIP_ERR_HIP = -102,      /* a HIP runtime call failed */
    LBFGS_HIP_ERR_COMM = -103,     /* RCCL / callback all-reduce failed */
    LBFGS_HIP_ERR_NOMEM = -104,
    LBFGS_HIP_ERR_NO_DEVICE = -105 /* no usable GPU: there is no CPU fallback */
};

typedef struct lbfgs_hip_ctx lbfgs_hip_ctx;         /* device, stream, board, communicator */
typedef struct lbfgs_hip_vec lbfgs_hip_vec;         /* this rank's shard of one n-vector    */
typedef struct lbfgs_hip_history lbfgs_hip_history; /* the m (s, y, ys, alpha) corrections  */

/* contiguous shard [offset, offset + n_local) of a global vector of n_global */
typedef struct lbfgs_hip_shard {
    int32_t rank, world;
    uint64_t n_global;
    uint64_t offset;
    uint64_t n_local;
} lbfgs_hip_shard;

/* how reductions are closed across ranks */
enum {
    LBFGS_HIP_COMM_NONE = 0,     /* world == 1 */
    LBFGS_HIP_COMM_RCCL = 1,     /* ncclAllReduce(ncclDouble, ncclSum) on the context's stream */
    LBFGS_HIP_COMM_CALLBACK = 2, /* host callback (e.g. torch.distributed/gloo); stages through pinned memory */
    LBFGS_HIP_COMM_P2P = 3       /* direct xGMI exchange: every rank stores its partial, as tagged 8-byte granules,
                                    into each peer's IPC-mapped mailbox and sums the P values in rank order --
                                    one tiny kernel per reduction, bitwise identical on every rank */
};
/* in-place sum all-reduce of `count` doubles in host memory; return 0 on success */
typedef int (*lbfgs_hip_allreduce_cb)(void* user, double* buf, int32_t count);

typedef struct lbfgs_hip_comm {
    int32_t kind;
    int32_t exclusive_device;    /* P2P, RCCL: non-zero = this rank has its GPU to itself (one process per GPU, the deployment this
                                    library is built for).  Only then may a kernel that occupies the WHOLE chip while it waits
                                    for its peers be used (the on-chip-resident two-loop, rust-lbfgs_amd/csrc/resident.h): ranks
                                    that share a GPU would keep each other from running.  0 = assume the GPU may be shared.
                                    (RCCL: only with LBFGS_HIP_RCCL_RESIDENT=1 in the environment as well -- opt-in, round 6: that form
                                    has never run with more than one rank.  The kernel's all-reduces then run gated on a second stream
                                    the context owns, and lbfgs_hip_ctx_create is collective -- same value, and same
                                    LBFGS_HIP_RCCL_RESIDENT, on every rank: the ranks vote on every rank's stream handshake and shard /
                                    device eligibility, try one such exchange under a chip-wide kernel on this communicator and vote
                                    on the outcome; if any rank fails, every rank runs the two-loop with a kernel per step and says so;
                                    a rank that fails locally in between still makes every collective and gets its error afterwards:
                                    rust-lbfgs_amd/csrc/ext_protocol.h.) */
    const void* rccl_unique_id;  /* 128 bytes from lbfgs_hip_rccl_unique_id() on rank 0, shared out of band */
    lbfgs_hip_allreduce_cb callback;
    void* callback_user;
    void* p2p_mailbox;           /* P2P: this rank's mailbox from lbfgs_hip_p2p_mailbox_create (lbfgs_hip_ctx_create takes
                                    ownership whether it succeeds or not: on failure it frees the mailbox itself) */
    const void* p2p_handles;     /* P2P: world x 64-byte IPC handles in rank order, gathered out of band */
    double p2p_timeout_s;        /* P2P: bound on every spin (0 = 5 s); a timeout fails the next scalars_read */
} lbfgs_hip_comm;

/* ------------------------------------------------------------------------- */
/* context                                                                     */
/* ------------------------------------------------------------------------- */
int lbfgs_hip_abi_version(void);
/* 16 hex digits: hash of the sources and compiler flags this library was built from (rust-lbfgs_amd/_build.py stamps
 * it; "unstamped" for a hand build).  The Python loader and tests/test_abi_exports.py compare it with the hash of the
 * checked-out sources, so a stale prebuilt library is rebuilt -- or refused -- instead of being trusted by its mtime. */
const char* lbfgs_hip_build_id(void);
int lbfgs_hip_device_count(int* count);
/* rank 0: produce the 128-byte RCCL unique id to broadcast to the other ranks */
int lbfgs_hip_rccl_unique_id(void* out128);
/* P2P step 1 (before ctx_create): allocate this rank's mailbox and export a 64-byte handle; the host framework
 * all-gathers the handles into lbfgs_hip_comm.p2p_handles.  Two placements:
 *   DEVICE  uncached device memory, exported as a HIP IPC handle and mapped by the peers over xGMI (the fast one);
 *   HOST    a POSIX shared-memory segment registered with HIP (host-coherent, reached over PCIe by every GPU): for
 *           machines where IPC mapping of device memory between the GPUs fails or is slow.  Same kernels, same bits.
 * lbfgs_hip_p2p_mailbox_create takes the placement from the environment (LBFGS_HIP_P2P_MAILBOX=device|host,
 * default device).  ctx_create recognises the placement of every peer's handle by itself (ranks need not agree). */
enum { LBFGS_HIP_MAILBOX_AUTO = -1, LBFGS_HIP_MAILBOX_DEVICE = 0, LBFGS_HIP_MAILBOX_HOST = 1 };
int lbfgs_hip_p2p_mailbox_create(int device, void** mailbox_out, void* ipc_handle64_out);
int lbfgs_hip_p2p_mailbox_create2(int device, int placement, void** mailbox_out, void* ipc_handle64_out);
void lbfgs_hip_p2p_mailbox_destroy(int device, void* mailbox); /* only for a mailbox never handed to lbfgs_hip_ctx_create */
/* After EVERY rank's lbfgs_hip_ctx_create has returned (a barrier of the host framework): removes the name of this
 * rank's HOST-placed mailbox from the system, so that the segment cannot outlive the processes that map it (it is
 * also removed by lbfgs_hip_ctx_destroy).  No-op for the DEVICE placement.  *placement_out (may be NULL) receives
 * the placement of this rank's mailbox, or -1 without a P2P communicator. */
int lbfgs_hip_ctx_p2p_seal(lbfgs_hip_ctx* ctx, int* placement_out);
/* shard == NULL => single rank holding n elements.  comm == NULL => NONE.
 * stream == NULL => the context creates its own non-blocking stream. */
int lbfgs_hip_ctx_create(lbfgs_hip_ctx** out, int device, uint64_t n, const lbfgs_hip_shard* shard,
                         const lbfgs_hip_comm* comm, void* stream);
void lbfgs_hip_ctx_destroy(lbfgs_hip_ctx* ctx);
const char* lbfgs_hip_last_error(const lbfgs_hip_ctx* ctx); /* ctx may be NULL: creation errors */
/* waits for the stream; returns the device error, if any, like lbfgs_hip_scalars_read (a resident two-loop that timed out
 * is re-run with a kernel per step first) */
int lbfgs_hip_sync(lbfgs_hip_ctx* ctx);
void* lbfgs_hip_stream(lbfgs_hip_ctx* ctx); /* hipStream_t */
int lbfgs_hip_get_shard(const lbfgs_hip_ctx* ctx, lbfgs_hip_shard* out);
/* What the communicator of this context really spans, and what its exchanges cost -- so that a multi-GPU measurement can be
 * cross-checked ("did RCCL see N ranks?") and a scaling shortfall attributed ("how long does one exchange take?").
 * Synchronises the stream. */
#define LBFGS_HIP_XCHG_BINS 48
typedef struct lbfgs_hip_comm_info {
    int32_t kind;               /* LBFGS_HIP_COMM_* */
    int32_t world, rank;        /* the shard this context was created with */
    int32_t ranks_seen;         /* RCCL: ncclCommCount of the communicator (ctx_create fails unless it equals world);
                                   P2P: mailboxes this rank can reach = its own + the peers' it mapped; none: 1; callback: 0
                                   (the library sees no peer itself) */
    int32_t rank_seen;          /* RCCL: ncclCommUserRank; otherwise rank */
    int32_t mailbox_placement;  /* P2P: LBFGS_HIP_MAILBOX_DEVICE / _HOST of this rank's own mailbox; otherwise -1 */
    int32_t peers_device, peers_host; /* P2P: peers' mailboxes mapped, by placement */
    int32_t exclusive_device;   /* lbfgs_hip_comm.exclusive_device as given */
    int32_t _pad;
    uint64_t two_loops;          /* two-loop recursions enqueued so far (any launch form) */
    uint64_t two_loop_exchanges; /* cross-rank exchanges enqueued inside them (world > 1; 2*bound or 2*bound+1 each) */
    uint64_t allreduce_launches; /* stand-alone all-reduces enqueued (RCCL / callback / the P2P kernel) */
    uint64_t p2p_exchanges;      /* P2P exchanges enqueued so far, in-kernel ones included */
    uint64_t resident_fallbacks; /* resident two-loops that timed out and were re-run with a kernel per step */
    /* measured ON THE DEVICE by the one workgroup that closes a reduction across ranks (P2P only; wall clock, 100 MHz),
     * class [0] = exchanges outside a two-loop, [1] = inside one: */
    uint64_t timed_exchanges[2]; /* exchanges the figures below cover */
    double exchange_us[2];       /* total time inside the exchange proper: stores to every peer's mailbox + wait for every
                                    peer's values (includes waiting for a peer that arrives later) */
    double local_wait_us[2];     /* total time that workgroup waited for this GPU's other workgroups before it */
    /* ... and their distribution (ABI 5), same classes: */
    double exchange_us_max[2];   /* the longest single exchange */
    double local_wait_us_max[2]; /* the longest single wait for this GPU's own workgroups */
    uint32_t exchange_hist[2][LBFGS_HIP_XCHG_BINS]; /* exchanges by duration: bin b < 32 covers [b/4, (b+1)/4) us; bin 32 + k covers
                                    [8 * 2^k, 8 * 2^(k+1)) us, the last one everything beyond (quantiles: rust-lbfgs_amd/api.py
                                    exchange_quantile; bench.py reports p50 / p99 per leg) */
} lbfgs_hip_comm_info;
int lbfgs_hip_ctx_comm_info(lbfgs_hip_ctx* ctx, lbfgs_hip_comm_info* out);
/* launch geometry override for tuning (0 = default): blocks, i.e. workgroups per launch */
int lbfgs_hip_set_grid(lbfgs_hip_ctx* ctx, int blocks);
/* which code path the two-loop recursions of this context took so far: how many ran as the single on-chip-resident
 * kernel (rust-lbfgs_amd/csrc/resident.h) rather than as one launch per step, and how many elements of this rank's
 * shard the last such launch kept on the chip (= n_local unless the shard is larger than the chip: "hybrid")
 * (diagnostics, tests, bench; either pointer may be NULL) */
int lbfgs_hip_path_stats(lbfgs_hip_ctx* ctx, uint64_t* resident_two_loops, uint64_t* resident_elements);

/* ------------------------------------------------------------------------- */
/* vectors: Vec<f64> of the reference (core.rs:24-39, lbfgs.rs:611-613)        */
/* ------------------------------------------------------------------------- */
int lbfgs_hip_vec_alloc(lbfgs_hip_ctx* ctx, lbfgs_hip_vec** out);          /* vec![0.0; n] */
void lbfgs_hip_vec_free(lbfgs_hip_vec* v);
int lbfgs_hip_vec_upload(lbfgs_hip_vec* v, const double* host, uint64_t count);   /* local shard */
int lbfgs_hip_vec_download(const lbfgs_hip_vec* v, double* host, uint64_t count); /* synchronises */
int lbfgs_hip_vec_fill(lbfgs_hip_vec* v, double value);
void* lbfgs_hip_vec_ptr(lbfgs_hip_vec* v);       /* device pointer of the shard (for device-side evaluators) */
int lbfgs_hip_vec_swap(lbfgs_hip_vec* a, lbfgs_hip_vec* b); /* O(1) exchange of the two buffers */

/* ------------------------------------------------------------------------- */
/* scalar board                                                                */
/* ------------------------------------------------------------------------- */
int lbfgs_hip_scalars_read(lbfgs_hip_ctx* ctx, int first, int count, double* host); /* synchronises */
int lbfgs_hip_scalars_write(lbfgs_hip_ctx* ctx, int first, int count, const double* host);
void* lbfgs_hip_scalars_ptr(lbfgs_hip_ctx* ctx);
/* sum all-reduce of board[first, first+count) across ranks (no-op when world == 1): closes a
 * reduction whose per-rank partial the HOST produced (a sharded host closure's partial f) */
int lbfgs_hip_scalars_allreduce(lbfgs_hip_ctx* ctx, int first, int count);

/* ------------------------------------------------------------------------- */
/* primitives: impl LbfgsMath<f64> for [f64]  (src/math.rs:31-82)              */
/* ------------------------------------------------------------------------- */
int lbfgs_hip_vecadd(lbfgs_hip_vec* y, const lbfgs_hip_vec* x, double c);        /* math.rs:33  y += c*x */
int lbfgs_hip_vecadd_dev(lbfgs_hip_vec* y, const lbfgs_hip_vec* x, int c_slot);  /* same, c read from the board */
int lbfgs_hip_vecdot(const lbfgs_hip_vec* x, const lbfgs_hip_vec* y, int out_slot); /* math.rs:40 */
int lbfgs_hip_vecscale(lbfgs_hip_vec* y, double c);                              /* math.rs:45 */
int lbfgs_hip_veccpy(lbfgs_hip_vec* y, const lbfgs_hip_vec* x);                  /* math.rs:52 */
int lbfgs_hip_vecncpy(lbfgs_hip_vec* y, const lbfgs_hip_vec* x);                 /* math.rs:59 */
int lbfgs_hip_vecdiff(lbfgs_hip_vec* z, const lbfgs_hip_vec* x, const lbfgs_hip_vec* y); /* math.rs:66 */
/* math.rs:73-81: the board receives the SQUARED norm (dot(x,x)); sqrt / 1.0/sqrt are host scalar ops */
int lbfgs_hip_vec2norm_sq(const lbfgs_hip_vec* x, int out_slot);
/* math.rs:73-76 vec2norm = sqrt(vecdot(x, x)) and math.rs:79-81 vec2norminv = 1.0 / vec2norm: the 1:1 forms of the
 * trait methods.  The (global, all-reduced) dot(x,x) is left in board[scratch_slot]; *out receives the norm resp.
 * its reciprocal (one stream synchronisation, like every host-returning reduction). */
int lbfgs_hip_vec2norm(const lbfgs_hip_vec* x, int scratch_slot, double* out);
int lbfgs_hip_vec2norminv(const lbfgs_hip_vec* x, int scratch_slot, double* out);

/* ------------------------------------------------------------------------- */
/* fused hot-path operators                                                    */
/* ------------------------------------------------------------------------- */

/* Problem::take_line_step (core.rs:155-164): x = xp + step*d, then, when wp != NULL,
 * Orthantwise::constraint_line_search (orthantwise.rs:118-133): x_i = 0 where
 * signum(x_i) != signum(wp_i) for global i in [start, end).  2r(+1r) 1w. */
int lbfgs_hip_line_step(lbfgs_hip_vec* x, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double step,
                        const lbfgs_hip_vec* wp, uint64_t start, uint64_t end);

/* Progress::new / Report::new norms (core.rs:261-262, 294-295) in one pass:
 * board[out_slot] = ||x||^2, board[out_slot+1] = ||g||^2.  2r. */
int lbfgs_hip_norms_sq(const lbfgs_hip_vec* x, const lbfgs_hip_vec* g, int out_slot);

/* history (lbfgs.rs:607-627 IterationData x m) */
int lbfgs_hip_history_create(lbfgs_hip_ctx* ctx, int m, lbfgs_hip_history** out);
void lbfgs_hip_history_destroy(lbfgs_hip_history* h);
lbfgs_hip_vec* lbfgs_hip_history_s(lbfgs_hip_history* h, int slot);
lbfgs_hip_vec* lbfgs_hip_history_y(lbfgs_hip_history* h, int slot);
/* read/write the per-slot scalars ys[m], alpha[m] (device arrays owned by the history) */
int lbfgs_hip_history_scalars_read(lbfgs_hip_history* h, double* ys, double* alpha);  /* synchronises */
int lbfgs_hip_history_scalars_write(lbfgs_hip_history* h, const double* ys, const double* alpha);

/* IterationData::update (lbfgs.rs:640-692), vector part, ONE pass (4r 2w):
 *   s = x - xp ; y = g - gp ;
 *   board[out_slot+0] = ||s||^2      (:645, squared)
 *   board[out_slot+1] = y.s          (:653)  also stored as ys[slot] (:656)
 *   board[out_slot+2] = y.y          (:654)
 *   board[out_slot+3] = ||x||^2, board[out_slot+4] = ||g||^2   (free: x and g are read anyway)
 *   board[out_slot+5] = s.bs with bs = gp*(-step)               (:670-673; 0 unless damping != 0)
 *   board[out_slot+6] = s.(-g): the numerator of the two-loop's first alpha (lbfgs.rs:587 with d = -g), free
 *                       here; pass the slot to lbfgs_hip_two_loop_from to skip that pass (not under OWL-QN,
 *                       where the direction starts from -pg)
 * The caller does the scalar tests (:646, :655) and gamma = ys/yy (:691).  */
int lbfgs_hip_history_update(lbfgs_hip_history* h, int slot, const lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                             const lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, double step, int damping,
                             int out_slot);
/* Powell damping case 1 (lbfgs.rs:675-680): y = (gp*(-step))*(1-theta) + theta*y.  2r 1w. */
int lbfgs_hip_history_damp(lbfgs_hip_history* h, int slot, const lbfgs_hip_vec* gp, double step, double theta);

/* update_search_direction + lbfgs_two_loop_recursion + dnorm
 * (core.rs:95-101, lbfgs.rs:569-604, lbfgs.rs:543), fused to 8*bound + 1 passes:
 *   d = -g ; two-loop over the history ; board[dnorm_slot] = ||d||^2 ; board[dnorm_slot+1] = g.d
 *   (the dginit of the NEXT line search, core.rs:78-92, free because the last step streams d anyway).
 * gamma is read on the device as board[gamma_num_slot] / board[gamma_den_slot]
 * (ys / yy of the update just done).  k and end are the reference's arguments
 * (k = number of corrections stored so far, end = slot just written); *new_end
 * receives (end+1)%m.  alpha[] of the history is overwritten (lbfgs.rs:587).
 * `g` is gx, or pg under OWL-QN (core.rs:96-97). */
int lbfgs_hip_two_loop(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                       int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int* new_end);
/* the same, starting from an alpha_0 numerator that is already on the board (first_dot_slot >= 0: the value
 * lbfgs_hip_history_update left at out_slot+6 for the slot `end` and this `g`): 8*bound - 1 passes */
int lbfgs_hip_two_loop_from(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                            int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end);
/* two-loop under OWL-QN with constrain_search_direction (orthantwise.rs:140-161, lbfgs.rs:554) folded into the
 * last step, which streams pg anyway:  board[dnorm_slot+0] = ||d||^2 BEFORE the projection (lbfgs.rs:543),
 * +2 = ||d||^2 after it (caller asserts != 0), +3 = pg.d after it (the next dginit, core.rs:90); +1 unused. */
int lbfgs_hip_two_loop_owlqn(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* pg, uint64_t k, int end,
                             int gamma_num_slot, int gamma_den_slot, int dnorm_slot, uint64_t start, uint64_t end_,
                             int* new_end);
/* EXTENSION: the same recursion in the coefficient space of the basis [s_j.., y_j.., g]
 * ("vector-free" L-BFGS): one pass refreshes the three changed rows of the Gram matrix B^T B, a scalar
 * kernel runs lbfgs.rs:582-601 on 2m+1 coefficients, one pass forms d.  4m+3 passes and 2 all-reduces
 * instead of 8m+1 and 2m+1.  Mathematically identical; rounding differs (tests bound it by 1e-10 against
 * the exact recursion).  Requires that it is called after EVERY history update (the Gram matrix is
 * maintained incrementally) and m <= 10.  Same arguments and outputs as lbfgs_hip_two_loop, plus two figures by which the
 * caller judges the coefficient-space arithmetic at run time (dnorm_slot must have 4 slots):
 *   board[dnorm_slot+2] = ||d||^2 as the Gram entries PREDICT it (delta^T G delta; NaN if a coefficient is not finite), next
 *                         to board[dnorm_slot] = ||d||^2 summed over the direction itself;
 *   board[dnorm_slot+3] = the CANCELLATION of the recursion's running vector in the basis, max over its two turning points of
 *                         sum_t ||b_t|| |delta_t| / ||sum_t delta_t b_t||: how many times more rounding error a dot product
 *                         taken through Gram entries carries than the exact recursion's (NaN: no digits left).
 * When they say that the Gram entries have lost the digits the recursion needs -- a run that blows up, one converged to
 * rounding level, more history vectors than dimensions -- the caller forms this direction again with
 * lbfgs_hip_two_loop_from (the solver: prediction off by > 1e-8, or cancellation > 1e4). */
int lbfgs_hip_two_loop_gram(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                            int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int* new_end);
/* the same recursion as the reference's UNFUSED sequence of primitives (10*bound+2 passes);
 * d must already hold -g.  Kept as the on-device cross-check of the fused kernels. */
int lbfgs_hip_two_loop_unfused(lbfgs_hip_history* h, lbfgs_hip_vec* d, uint64_t k, int end,
                               int gamma_num_slot, int gamma_den_slot, int* new_end);

/* ------------------------------------------------------------------------- */
/* OWL-QN operators (src/orthantwise.rs, core.rs:119-132,167-180,213-217)      */
/* [start, end) are GLOBAL indices, already clamped by the caller (orthantwise.rs:59-67) */
/* ------------------------------------------------------------------------- */
/* Problem::evaluate tail (core.rs:123-126): board[out_slot] = sum c*|x_i| over [start,end)
 * (orthantwise.rs:70-79), pg = pseudo-gradient (orthantwise.rs:82-112),
 * board[out_slot+1] = ||pg||^2 (core.rs:185), board[out_slot+2] = ||x||^2.  2r 1w. */
int lbfgs_hip_owlqn_post_eval(const lbfgs_hip_vec* x, const lbfgs_hip_vec* g, lbfgs_hip_vec* pg, double c,
                              uint64_t start, uint64_t end, int out_slot);
/* Problem::update_orthant_new_point (core.rs:167-180): wp_i = xp_i==0 ? signum(-pg_i) : signum(xp_i), all i */
int lbfgs_hip_orthant_select(lbfgs_hip_vec* wp, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* pg);
/* Orthantwise::constrain_search_direction (orthantwise.rs:140-161): d_i = 0 where
 * signum(d_i) != signum(-pg_i) on [start,end); board[out_slot] = ||d||^2 (caller asserts != 0);
 * board[out_slot+1] = pg.d, the dginit of the next line search under OWL-QN (core.rs:90). */
int lbfgs_hip_constrain_direction(lbfgs_hip_vec* d, const lbfgs_hip_vec* pg, uint64_t start, uint64_t end,
                                  int out_slot);

/* ------------------------------------------------------------------------- */
/* device-resident objectives (the "user closure" of core.rs:119-121 kept in HBM) */
/* ------------------------------------------------------------------------- */
enum {
    LBFGS_HIP_OBJ_QUADRATIC = 1,  /* hashed diagonal quadratic  (BASELINE.json configs 2, 4) */
    LBFGS_HIP_OBJ_LOGISTIC = 2,   /* hashed separable logistic  (config 3, with OWL-QN)      */
    LBFGS_HIP_OBJ_ROSENBROCK = 3, /* src/lib.rs:79-94 default_evaluate (pairs must not straddle shards) */
    LBFGS_HIP_OBJ_LJ_ALLPAIRS = 4,  /* examples/lj.rs:20-64,113-118: exact all-pairs Lennard-Jones, x = 3*natoms; one rank */
    LBFGS_HIP_OBJ_LJ_NEIGHBORS = 5, /* the same pair terms over a fixed, caller-built neighbour table with a cutoff
                                       (shifted by v(rc)); the table is never rebuilt: valid only while it stays complete */
    LBFGS_HIP_OBJ_LJ_CELLS = 6      /* E = sum_{i<j, r<rc} [v(r) - v(rc)] through a neighbour (Verlet) list that the LIBRARY
                                       builds on the device from a cell list (radius cutoff + skin) and rebuilds whenever an
                                       atom has moved more than skin/2 since the last build -- detected inside the
                                       evaluation kernel, followed by rebuild + re-evaluation, so every result is the exact
                                       cutoff sum at x: the evaluator of BASELINE config 5 at 1e6 atoms; one rank */
};
typedef struct lbfgs_hip_objective {
    int32_t kind;
    int32_t _pad;
    uint64_t seed_a, seed_b;
    /* LJ_NEIGHBORS: ELL table int32[max_nbr][natoms] in DEVICE memory (column-major, -1 = empty), from
     * lbfgs_hip_device_buffer_create; cutoff rc */
    const void* nbr_index;
    uint32_t max_nbr;   /* LJ_NEIGHBORS: rows of the table.  LJ_CELLS: capacity of an atom's list (rounded up to a multiple
                           of 16; 0 = 128); an atom with more neighbours within cutoff + skin fails the evaluation */
    uint32_t _pad2;
    double cutoff;
    double skin;        /* LJ_CELLS: list radius = cutoff + skin (> 0) */
} lbfgs_hip_objective;
/* LJ_CELLS bookkeeping of this context: list builds so far, evaluations (re-evaluations after a rebuild included),
 * and the longest neighbour list of the latest build */
int lbfgs_hip_lj_cells_stats(lbfgs_hip_ctx* ctx, uint64_t* rebuilds, uint64_t* evaluations, uint32_t* longest_list);
/* page-locked host staging memory (hipHostMalloc): the host-closure bridge keeps x and g in such buffers so that the two
 * PCIe transfers per evaluation run at DMA speed instead of through pageable memory */
int lbfgs_hip_host_buffer_create(lbfgs_hip_ctx* ctx, uint64_t bytes, void** out);
void lbfgs_hip_host_buffer_destroy(lbfgs_hip_ctx* ctx, void* buf);
/* raw device buffers for objective data (neighbour tables ...): allocate + upload; free */
int lbfgs_hip_device_buffer_create(lbfgs_hip_ctx* ctx, const void* host, uint64_t bytes, void** out);
void lbfgs_hip_device_buffer_destroy(lbfgs_hip_ctx* ctx, void* buf);
/* g = grad f(x); board[out_slot] = f(x) (global sum).  1r 1w. */
int lbfgs_hip_objective_eval(const lbfgs_hip_objective* obj, const lbfgs_hip_vec* x, lbfgs_hip_vec* g,
                             int out_slot);
/* take_line_step + evaluate + dg_unchecked in ONE pass (core.rs:155-158,119-121,114-116), no OWL-QN:
 *   x = xp + step*d ; g = grad f(x) ; board[out_slot] = f(x) ; board[out_slot+1] = g.d.   2r 2w. */
int lbfgs_hip_objective_line_eval(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                  const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* g, int out_slot);

/* Deferred trial points, for objectives that are evaluated element by element (QUADRATIC, LOGISTIC).
 * A line search reads two scalars from each trial (line.rs:283-288: f and g.d); the vectors x and g of a trial
 * are only used if it is the LAST one (by IterationData::update, lbfgs.rs:525).  So:
 *   lbfgs_hip_objective_line_probe      f(xp + step*d) and grad.d, nothing written.                 2r 0w
 *   lbfgs_hip_history_update_from_step  x = xp + t*d ; g = grad f(x) ; then lbfgs_hip_history_update's
 *                                       s, y and seven sums (same board layout) in the same pass.  3r 4w
 * A search with T trials + the update moves 2T + 7 passes instead of 4T + 6.  The arithmetic per element is that of
 * lbfgs_hip_objective_line_eval followed by lbfgs_hip_history_update (s = (xp + t*d) - xp, not t*d).
 * lbfgs_hip_objective_is_elementwise: 1 if `obj` supports the two calls, else 0 (they return LBFGS_HIP_ERR_ARG). */
int lbfgs_hip_objective_is_elementwise(const lbfgs_hip_objective* obj);
int lbfgs_hip_objective_line_probe(const lbfgs_hip_objective* obj, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d,
                                   double step, int out_slot);
int lbfgs_hip_history_update_from_step(lbfgs_hip_history* h, int slot, const lbfgs_hip_objective* obj,
                                       lbfgs_hip_vec* x, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double t,
                                       lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, double step, int damping,
                                       int out_slot);

/* one OWL-QN line-search trial in ONE pass (line.rs:740-743): take_line_step with the orthant projection
 * (core.rs:155-164), evaluate (core.rs:119-126) incl. x1norm and the pseudo-gradient, dg_unchecked:
 *   board[out_slot+0] = f(x), +1 = g.d, +2 = sum c*|x| on [start,end), +3 = ||pg||^2, +4 = ||x||^2.   3r 3w */
int lbfgs_hip_objective_owlqn_line_eval(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                        const lbfgs_hip_vec* d, double step, const lbfgs_hip_vec* wp, lbfgs_hip_vec* g,
                                        lbfgs_hip_vec* pg, double c, uint64_t start, uint64_t end, int out_slot);

/* The FIRST trial of an OWL-QN search, with Problem::update_orthant_new_point (line.rs:735, core.rs:167-180) folded in: the
 * orthant wp_i = xp_i == 0 ? signum(-pg_i) : signum(xp_i) is formed from xp -- which the trial streams anyway -- and the
 * pseudo-gradient of the point the search starts from (pg on entry), written to wp for the later trials of the search
 * (lbfgs_hip_objective_owlqn_line_eval), and used at once; pg then receives the trial point's pseudo-gradient, as there.
 * Same sums, same board layout.  3r 4w, where lbfgs_hip_orthant_select (2r 1w) + the trial (3r 3w) move 5r 4w. */
int lbfgs_hip_objective_owlqn_first_trial(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                          const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* wp, lbfgs_hip_vec* g,
                                          lbfgs_hip_vec* pg, double c, uint64_t start, uint64_t end, int out_slot);

/* An OWL-QN trial that ALSO does IterationData::update for its point (lbfgs.rs:640-656), speculatively (ABI 5): next to what
 * lbfgs_hip_objective_owlqn_line_eval / _first_trial (first != 0) leave, s = x - xp and y = g - gp go into history slot `slot`
 * (the slot the update is about to fill), ys[slot] is set, and board[upd_slot+0] = ||s||^2, +1 = y.s, +2 = y.y -- the first
 * three sums of lbfgs_hip_history_update, which the caller then skips.  A backtracking search accepts the trial it evaluated
 * LAST (line.rs:747-777), so the last trial's s, y and sums are the update's; a search that failed and was reverted
 * (line.rs:213-220) must run lbfgs_hip_history_update itself (x == xp then: "x not changed", lbfgs.rs:646).  4r 6w (first: 4r 7w)
 * against 3r 3w + 4r 2w.  Not for Powell damping (lbfgs.rs:664-689 needs s.bs as well).  Objectives without a fused form run
 * the separate passes (upd_slot then needs the 7 slots of lbfgs_hip_history_update). */
int lbfgs_hip_objective_owlqn_trial_update(const lbfgs_hip_objective* obj, lbfgs_hip_history* h, int slot, lbfgs_hip_vec* x,
                                           const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* wp,
                                           int first, lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, lbfgs_hip_vec* pg, double c,
                                           uint64_t start, uint64_t end, int out_slot, int upd_slot);

/* ------------------------------------------------------------------------- */
/* measurement                                                                 */
/* ------------------------------------------------------------------------- */
/* Kernel classes timed with HIP events on the context's stream when profiling is on. */
enum {
    LBFGS_HIP_K_TWOLOOP_STEP = 0, /* q += c*u ; out = v.q  (3r 1w) -- the dominant kernel */
    LBFGS_HIP_K_TWOLOOP_EDGE = 1, /* first dot (2r), gamma transition and last step (2r 1w) */
    LBFGS_HIP_K_UPDATE = 2,
    LBFGS_HIP_K_LINE = 3,
    LBFGS_HIP_K_EVAL = 4,
    LBFGS_HIP_K_OWLQN = 5,
    LBFGS_HIP_K_BLAS1 = 6,
    LBFGS_HIP_K_COMM = 7,
    LBFGS_HIP_K_TWOLOOP_ALL = 8,  /* one event pair around a whole lbfgs_hip_two_loop call */
    LBFGS_HIP_K_TWOLOOP_RESIDENT = 9, /* the whole recursion as ONE kernel, q kept on chip (shards <= ~1.25e7 elements) */
    LBFGS_HIP_K_CLASSES = 10
};
int lbfgs_hip_prof_enable(lbfgs_hip_ctx* ctx, int on);
int lbfgs_hip_prof_reset(lbfgs_hip_ctx* ctx);
/* synchronises; launches and summed milliseconds of a class since the last reset */
int lbfgs_hip_prof_read(lbfgs_hip_ctx* ctx, int kclass, uint64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif
