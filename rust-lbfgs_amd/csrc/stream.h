// rust-lbfgs_amd/csrc/stream.h -- the one streaming skeleton every hot-path kernel is built on.
//
// All L-BFGS / OWL-QN vector work is BLAS-1: each element is touched O(1) times,
// arithmetic intensity ~0.1-0.25 flop/byte, so the bound is HBM bandwidth and
// nothing here uses MFMA.  What matters on MI355X (gfx950):
//   * 16-byte-per-lane coalesced loads/stores (double2 -> global_load_dwordx4):
//     one wave instruction moves 1 KiB;
//   * enough bytes in flight per CU: every thread issues UNROLL independent
//     16-B loads per input stream before the first use (compiler places the
//     s_waitcnt at first use);
//   * FEW, LONG DRAM streams: a persistent grid of ~0.85 workgroups per CU sweeps
//     each vector as one contiguous window (grid-stride over 4-16 KiB chunks).
//     Measured on MI355X at n = 1e8 (tools/tune_stream.hip): one contiguous range
//     per workgroup with 8 workgroups/CU = 5.2 TB/s on the 3r+1w two-loop step;
//     the windowed map with 216 workgroups and `nt` hints = 6.3 TB/s; pure reads
//     7.2 TB/s.  More workgroups mean more concurrent DRAM pages, not more speed;
//   * the `nt` (streaming) cache hint on every load/store once a vector no longer
//     fits the 256 MiB Infinity Cache; below that, plain accesses let the cache
//     keep the running vector between consecutive kernels;
//   * reductions: f64 per-thread accumulators -> wave64 shuffle tree -> LDS ->
//     one partial per workgroup, published as tagged 8-byte granules -> the
//     last-index workgroup polls them and sums in a FIXED order.  No float
//     atomics, so results are bitwise reproducible for a given (n, grid).
//     A sum that only the NEXT kernel consumes is not reduced at all by its
//     producer: the consumer adds the partials in its prologue (sum_partials);
//   * what a DEPENDENT kernel costs besides its bytes (the bound below ~100 MB
//     per vector): no release fences towards the host (a system-scope release
//     writes back a whole L2), no scratch memory (+12 us per dispatch), first
//     loads before the board reads, launch-to-launch sequence numbers in device
//     memory (DevCounters) instead of kernel arguments.
//
// An operator (ops.h) only states: its input/output streams, how one element is
// computed, and how many sums it accumulates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lh {

#ifndef LH_BLOCK         // overridable for in-situ A/B builds (tools/build_variants.sh)
#define LH_BLOCK 256
#endif
constexpr int BLOCK = LH_BLOCK;     // 4 wave64 per workgroup
constexpr int WAVES = BLOCK / 64;
constexpr int UNROLL = 4;           // default: independent 16-B loads per stream per thread
constexpr int DEFAULT_MAP = 2;      // default address map (see stream_kernel)
constexpr int MAX_RED = 64;         // sums a kernel may produce (the Gram rows need 3*(2m+1))
constexpr int MAX_GRID = 4096;      // upper bound on workgroups per launch

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int RED_PTRS = 8;         // sums that may go to individually addressed slots

// ---- direct xGMI exchange of reduction results (communicator kind "p2p") ---------------------------
// xGMI is point-to-point and fully connected and the messages are 8-48 bytes, so instead of a collective
// launch per reduction the LAST workgroup of the producing kernel stores this rank's sums straight into
// every peer's mailbox and adds up what the peers stored into its own.  A double travels as two 8-byte
// granules {tag = epoch, 32 data bits} (an 8-byte store is atomic, so data and flag arrive together: the
// LL-protocol idea).  Mailboxes are uncached device memory mapped through IPC handles; all accesses are
// system-scope atomics.  The sum runs in rank order on every rank => bitwise identical results everywhere.
// Every spin is bounded by a wall-clock timeout that raises *err.
constexpr int P2P_MAX_WORLD = 8;
constexpr int P2P_RING = 4;  // a rank can be at most one reduction ahead of a peer; 4 slots is ample
constexpr size_t P2P_MBOX_WORDS = (size_t)P2P_RING * P2P_MAX_WORLD * MAX_RED * 2;
struct P2PCtl {
    unsigned long long* mbox[P2P_MAX_WORLD];  // [rank] -> that rank's mailbox (own entry = local)
    int world, rank;                          // world <= 1: no exchange
    unsigned int* err;
    unsigned long long timeout_ticks;         // wall_clock64 ticks (100 MHz)
    // GATED exchange (RCCL under the persistent two-loop kernel, resident.h): the reduction across ranks is closed OUTSIDE
    // the kernel, by ncclAllReduce launches the host has enqueued on a second stream behind gate kernels -- the kernel stores
    // its sums into ext_buf (uncached), raises flag A, and reads the reduced sums back once flag B carries the same epoch.
    double* ext_buf;                          // nullptr: the mailbox exchange above.  [EXT_SLOTS][EXT_SLOT_DOUBLES]
    unsigned long long* ext_flags;            // uncached: [0] = A (kernel -> gate), [1] = B (post -> kernel), [2] = abort
};
constexpr int EXT_SLOTS = 64, EXT_SLOT_DOUBLES = 8;  // a ring of exchange buffers: one per exchange of a launch (<= 2*24 + 1)
constexpr size_t EXT_BUF_OFFSET = 256;               // bytes from the start of the context's uncached block (flags) to that ring
__device__ __forceinline__ bool p2p_active(const P2PCtl& c) { return c.world > 1 || c.ext_buf != nullptr; }

// Host mirror of the scalar board: when the totals are FINAL (one rank, or closed by the in-kernel exchange)
// the last workgroup also stores them into host-mapped memory and then bumps a sequence word, so the host
// reads results with a short spin instead of a copy kernel + stream synchronisation.
struct MirrorCtl {
    double* host_board;             // device address of the host-mapped mirror (nullptr = off)
    const double* board;            // the device board (to turn an output pointer into a slot index)
    unsigned long long* host_seq;   // device address of the host-mapped sequence word (value: DevCounters::mirror_seq)
    unsigned long long* host_err;   // device address of the host-mapped copy of the P2P timeout flag
    int slots;                      // mirrored slots
};

// Launch-to-launch sequence numbers live in DEVICE memory, not in kernel arguments: every workgroup reads them when it
// starts and the reducer (the last workgroup) advances them before the kernel ends -- stream order makes the next kernel
// see the new values.  A kernel's arguments are then the same every time it is launched with the same operands, which
// keeps launches free of host-computed sequence numbers (round 2 replayed whole recursions as hipGraphs on the strength of it).  The host keeps a shadow of each counter.
struct DevCounters {
    unsigned int red_epoch;         // tag of the NEXT tagged hand-off: never 0, distinct from every tag still in `gran`
    unsigned int p2p_epoch;         // epoch of the NEXT P2P exchange: identical on all ranks, never 0
    unsigned long long mirror_seq;  // sequence number of the LATEST mirrored launch
};
__device__ __forceinline__ unsigned int next_epoch(unsigned int e) { return (e + 1u == 0u) ? 1u : e + 1u; }
// Run-time figures of the cross-rank exchange (world > 1 only), kept in the same device page as the counters, 64 bytes
// in: what the ONE workgroup that closes a reduction across ranks spent doing so, in wall-clock ticks (100 MHz).  Two
// classes: [0] every exchange outside a two-loop recursion, [1] the exchanges inside one (the 2*bound dependent ones the
// 8-GPU run's latency budget is made of).  Kernels run one after the other on the context's stream and one workgroup per
// kernel updates them, so a plain read-modify-write suffices.  lbfgs_hip_ctx_comm_info reads them.
struct DevXchg {
    unsigned long long p2p_ticks;    // inside p2p_exchange: publish to every peer + wait for every peer's values
    unsigned long long local_ticks;  // before that: the exchanging workgroup waiting for this GPU's other workgroups
    unsigned long long count;        // exchanges
    unsigned long long _pad;
};
constexpr int DEV_XCHG_OFFSET = 64;  // bytes from the start of the counters' page
// ... and the DISTRIBUTION of those exchanges, per class, 128 bytes in: how long the longest one took, the longest wait for this
// GPU's own workgroups, and a histogram of the exchange times (lbfgs_hip_comm_info::exchange_hist: quarter-microsecond bins up
// to 8 us, octaves beyond).  Updated by fire-and-forget atomics of the exchanging thread -- no value comes back, so nothing
// waits on them: the record above is loaded BEFORE an exchange for the same reason.
constexpr int XCHG_BINS = 48;
struct DevXchgDist {
    unsigned long long p2p_max, local_max;
    unsigned int hist[XCHG_BINS];
};
constexpr int DEV_XCHG_DIST_OFFSET = 128;
static_assert(DEV_XCHG_OFFSET + 2 * sizeof(DevXchg) <= DEV_XCHG_DIST_OFFSET && DEV_XCHG_DIST_OFFSET + 2 * sizeof(DevXchgDist) <= 4096,
              "the counters' page");
__device__ __forceinline__ void xchg_dist_note(DevCounters* ctr, const unsigned int cls, const unsigned long long p2p_ticks,
                                               const unsigned long long local_ticks) {
    DevXchgDist* const d = reinterpret_cast<DevXchgDist*>(reinterpret_cast<char*>(ctr) + DEV_XCHG_DIST_OFFSET) + (cls & 1u);
    unsigned int b;
    if (p2p_ticks < 800ull) {
        b = (unsigned int)p2p_ticks / 25u;  // 100 MHz ticks: 25 = 0.25 us
    } else {
        const unsigned long long oct = p2p_ticks / 800ull;  // >= 1
        b = 32u + (unsigned int)(63 - __clzll((long long)oct));
        if (b > (unsigned int)XCHG_BINS - 1u) b = (unsigned int)XCHG_BINS - 1u;
    }
    (void)__hip_atomic_fetch_add(&d->hist[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_max(&d->p2p_max, p2p_ticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_max(&d->local_max, local_ticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ DevXchg* dev_xchg(DevCounters* ctr, const unsigned int cls) {
    return reinterpret_cast<DevXchg*>(reinterpret_cast<char*>(ctr) + DEV_XCHG_OFFSET) + (cls & 1u);
}

struct RedCtl {
    double* partials;        // [MAX_RED][MAX_GRID] workgroup partial sums   (ticket hand-off: more than RED_PTRS sums)
    unsigned int* ticket;    // arrival counter, self-resetting
    unsigned long long* gran;     // [MAX_RED][MAX_GRID][2] tagged granules  (tagged hand-off: up to RED_PTRS sums)
    DevCounters* ctr;             // the context's counters
    unsigned int tagged;          // 1 = tagged hand-off (tag = ctr->red_epoch), 0 = the ticket form,
                                  // 2 = PARTIALS ONLY: every workgroup stores its partial sums to out[k][blockIdx.x] and
                                  //     exits; nobody waits.  The consumer -- the next kernel on the stream -- adds them up
                                  //     itself in its prologue (sum_partials below), in the reducer's order.
    unsigned int* err;            // device error flag (2 = a partial never arrived)
    unsigned int xchg_class;      // world > 1: which DevXchg record this launch's exchange is charged to (1 = inside a two-loop)
    unsigned long long timeout_ticks;  // bound on the reducer's spin (wall_clock64 ticks, 100 MHz)
    double* out[RED_PTRS];   // where the last workgroup puts the totals (NRED <= RED_PTRS) ...
    double* out_contig;      // ... or one contiguous array of NRED doubles (NRED > RED_PTRS)
    double* dup_ptr;         // optional second destination of total number dup_k (ys[slot] of the history)
    int dup_k;
    P2PCtl p2p;
    MirrorCtl mirror;
};

__device__ __forceinline__ size_t p2p_word(unsigned epoch, int src_rank, int k, int half) {
    return (((size_t)(epoch % P2P_RING) * P2P_MAX_WORLD + src_rank) * MAX_RED + k) * 2 + half;
}

// Called by ALL threads of one workgroup (>= 64 threads).  vals[0..count) in LDS: in = this rank's sums,
// out = the global sums.  bits: LDS scratch [P2P_MAX_WORLD][MAX_RED][2].
// `epoch` = this exchange's sequence number (DevCounters::p2p_epoch as read in the kernel's prologue).  It is a separate
// argument on purpose: a local copy of P2PCtl with the epoch patched in would be indexed dynamically (mbox[p]) and
// land in scratch memory -- 176 bytes per lane and +12 us of dispatch cost on EVERY reducing kernel (measured).
// The two halves of the exchange (resident.h calls them from different workgroups):
//   p2p_publish  store this rank's vals[0..count) into every rank's mailbox (own included); the first wave does it
//   p2p_collect  wait for all ranks' values of this epoch in the OWN mailbox and add them up in rank order -> vals
__device__ __forceinline__ void p2p_publish(const P2PCtl& c, const unsigned int epoch, const double* vals, int count) {
    const int per_rank = count * 2, total = c.world * per_rank;
    if (threadIdx.x < 64) {
        for (int i = threadIdx.x; i < total; i += 64) {
            const int p = i / per_rank, k = (i % per_rank) >> 1, h = i & 1;
            const unsigned long long b = (unsigned long long)__double_as_longlong(vals[k]);
            const unsigned int data = h ? (unsigned int)(b >> 32) : (unsigned int)b;
            __hip_atomic_store(c.mbox[p] + p2p_word(epoch, c.rank, k, h), ((unsigned long long)epoch << 32) | data,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__device__ __forceinline__ void p2p_collect(const P2PCtl& c, const unsigned int epoch, double* vals, int count,
                                            unsigned int (*bits)[MAX_RED][2], const unsigned long long extra_ticks = 0) {
    const int per_rank = count * 2, total = c.world * per_rank;
    if (threadIdx.x < 64) {
        const long long t0 = wall_clock64();
        for (int i = threadIdx.x; i < total; i += 64) {  // what rank r stored for me
            const int r = i / per_rank, k = (i % per_rank) >> 1, h = i & 1;
            const unsigned long long* src = c.mbox[c.rank] + p2p_word(epoch, r, k, h);
            unsigned long long g;
            for (;;) {
                g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned int)(g >> 32) == epoch) break;
                if ((unsigned long long)(wall_clock64() - t0) > c.timeout_ticks + extra_ticks) {
                    atomicExch(c.err, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            bits[r][k][h] = (unsigned int)g;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < count) {  // fixed-order sum over ranks
        double sum = 0.0;
        for (int r = 0; r < c.world; ++r) {
            const unsigned long long b = ((unsigned long long)bits[r][threadIdx.x][1] << 32) | bits[r][threadIdx.x][0];
            sum += __longlong_as_double((long long)b);
        }
        vals[threadIdx.x] = sum;
    }
    __syncthreads();
}
__device__ __forceinline__ void p2p_exchange(const P2PCtl& c, const unsigned int epoch, double* vals, int count,
                                             unsigned int (*bits)[MAX_RED][2]) {
    p2p_publish(c, epoch, vals, count);
    p2p_collect(c, epoch, vals, count, bits);
}
// The gated form (P2PCtl::ext_buf): called by ALL threads of one workgroup; vals[0..count) in LDS, in = this rank's sums, out =
// the global sums.  Slot epoch % EXT_SLOTS of the ring is the send AND receive buffer of the all-reduce the host enqueued for
// this epoch (in place).  Uncached memory, system-scope accesses: the values must have left the CU before the flag does
// (s_waitcnt, no L2 write-back: nothing here is cached), and the kernel that reduces them starts behind a kernel boundary.
__device__ __forceinline__ void ext_exchange(const P2PCtl& c, const unsigned int epoch, double* vals, int count) {
    double* const slot = c.ext_buf + (size_t)(epoch % (unsigned int)EXT_SLOTS) * EXT_SLOT_DOUBLES;
    if ((int)threadIdx.x < count) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + threadIdx.x), (unsigned long long)__double_as_longlong(vals[threadIdx.x]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(c.ext_flags, (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(c.ext_flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)epoch) {
            if ((unsigned long long)(wall_clock64() - t0) > c.timeout_ticks) {
                atomicExch(c.err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < count)
        vals[threadIdx.x] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(slot + threadIdx.x),
                                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    __syncthreads();
}
// The kernels of the second stream (host: lbfgs_hip.hip enqueue_gated_chain).  post: flag B = post_epoch (0: none) -- the
// all-reduce in front of it has completed (stream order).  gate: wait until the persistent kernel has raised flag A to
// gate_epoch (0: none) -- only then may the all-reduce behind it read the slot.  Bounded: a timeout or the abort word (the host
// gave the launch up) ends the wait; whatever is reduced then is discarded with the launch.
struct ExtGateArgs {
    unsigned long long* flags;
    unsigned int* err;
    unsigned long long post_epoch, gate_epoch, abort_id, timeout_ticks;
};
// (the kernel itself: lbfgs_hip.hip ext_post_gate_kernel -- this header is included by three translation units)

// ---- agent-scope accesses for the cross-workgroup hand-off -------------------------------
// Two forms, both built on 8-byte agent-scope atomics on both sides (no L2 write-back fence):
//  * tagged (every hot kernel: up to RED_PTRS sums).  A workgroup publishes each partial as two granules
//    {tag = launch epoch, 32 data bits} and EXITS; the last-index workgroup (dispatched last, so every other one is
//    already running or done) polls the G x NRED x 2 granules until all tags match and sums them in a fixed order.
//    Data and flag travel in one store, so there is no store drain, no ticket and no reload behind the ticket:
//    measured 6.9 -> 4.0 us per dependent reducing kernel at n = 1e5, 9.2 -> 6.3 us at n = 1e6
//    (profiles/r01_tune_tagged_epilogue.log).  The spin is bounded (timeout raises *err).
//  * ticket (more than RED_PTRS sums; also gram.h): partials stored write-through, drained (s_waitcnt vmcnt(0)),
//    then an arrival counter; the LAST ARRIVING workgroup reloads all partials.
// Both sum the partials in the same order, so results are bitwise independent of the form.
__device__ __forceinline__ void store_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p) {
    return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

// The same total without a trip through the LDS crossbar per level: rows of 16 lanes are folded with DPP moves
// (row_shl 8, 4, 2, 1: lane i takes lane i+k of its own row -- register-to-register, no ds_bpermute round trip), the four
// row sums are read with v_readlane and added in row order.  Order: each row a binary tree with strides 8, 4, 2, 1; then
// ((row0 + row1) + row2) + row3 -- a FIXED order, but not wave_sum's, so the two are not interchangeable where bits are
// compared.  The result is wave-uniform (valid in every lane).  ~0.1 us against ~0.3 us for wave_sum (six dependent
// ds_bpermute pairs); used where a reduction sits on a latency-critical path (resident.h's hand-off).
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(const double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);  // (bound_ctrl: a source lane outside the row reads as 0)
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(const double v, const int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov_f64<0x108>(v);  // row_shl:8
    v += dpp_mov_f64<0x104>(v);  // row_shl:4
    v += dpp_mov_f64<0x102>(v);  // row_shl:2
    v += dpp_mov_f64<0x101>(v);  // row_shl:1
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return ((r0 + r1) + r2) + r3;
}

// Sum `v` over the workgroup in a fixed order; result valid in thread 0.
template <int NRED>
__device__ __forceinline__ void block_sum(double (&acc)[NRED], double (*lds)[WAVES]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
        double w = wave_sum(acc[k]);
        if (lane == 0) lds[k][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) {
            double t = lds[k][0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) t += lds[k][w];
            acc[k] = t;
        }
    }
}

// the counters as every workgroup sees them when it STARTS: one 16-byte load issued in the kernel's prologue, so its
// latency hides behind the sweep instead of sitting in every workgroup's epilogue
__device__ __forceinline__ DevCounters load_counters(const RedCtl& red) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 v = *reinterpret_cast<const u4*>(red.ctr);
    DevCounters c;
    c.red_epoch = v.x;
    c.p2p_epoch = v.y;
    c.mirror_seq = ((unsigned long long)v.w << 32) | v.z;
    return c;
}

// The total of G workgroup partials, formed by EVERY workgroup of the consuming kernel in exactly the order in which the
// reducer of grid_reduce forms it (thread-strided partials, wave tree, waves in order), so that a sum closed this way is
// bitwise the sum the producing kernel's reducer would have stored.  All threads of the workgroup call it.
__device__ __forceinline__ double sum_partials(const double* parts, const unsigned int G) {
    __shared__ double lds[1][WAVES];
    __shared__ double total;
    double t[1] = {0.0};
    for (unsigned int b = threadIdx.x; b < G; b += BLOCK) t[0] += parts[b];
    block_sum<1>(t, lds);
    if (threadIdx.x == 0) total = t[0];
    __syncthreads();
    return total;
}

template <int NRED>
__device__ __forceinline__ void grid_reduce(double (&acc)[NRED], const RedCtl& red, const DevCounters& c0) {
    __shared__ double lds[NRED][WAVES];
    block_sum<NRED>(acc, lds);
    if constexpr (NRED <= RED_PTRS) {
        if (red.tagged == 2u) {  // partials only (uniform over the grid): no reducer, no waiting
            if (threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < NRED; ++k) red.out[k][blockIdx.x] = acc[k];
            }
            return;
        }
    }
    const unsigned int G = gridDim.x;
    double tot[NRED];
    const bool tagged = NRED <= RED_PTRS && red.tagged == 1u;  // (uniform over the grid)
    const unsigned int epoch = c0.red_epoch;   // (the reducer advances the counter only after every partial has arrived)
    long long t0 = 0;                          // when the reducer started waiting for the others (tagged form)
    if (tagged) {
        // ---- tagged hand-off ----
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < NRED; ++k) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(acc[k]);
                unsigned long long* g = red.gran + ((size_t)k * MAX_GRID + blockIdx.x) * 2;
                const unsigned long long tag = (unsigned long long)epoch << 32;
                __hip_atomic_store(g, tag | (b & 0xffffffffULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g + 1, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (blockIdx.x != G - 1u) return;  // uniform per workgroup
        __syncthreads();                   // lds is reused below
        t0 = wall_clock64();
#pragma unroll
        for (int k = 0; k < NRED; ++k) tot[k] = 0.0;
        for (unsigned int b = threadIdx.x; b < G; b += BLOCK) {  // same order as the ticket form: b strided by BLOCK
            unsigned long long lo[NRED], hi[NRED];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NRED; ++k) {  // 2*NRED independent loads in flight
                    const unsigned long long* g = red.gran + ((size_t)k * MAX_GRID + b) * 2;
                    lo[k] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hi[k] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < NRED; ++k)
                    ok = ok && (unsigned int)(lo[k] >> 32) == epoch && (unsigned int)(hi[k] >> 32) == epoch;
                if (ok) break;
                if ((unsigned long long)(wall_clock64() - t0) > red.timeout_ticks) {
                    atomicExch(red.err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < NRED; ++k)
                tot[k] += __longlong_as_double((long long)((hi[k] << 32) | (lo[k] & 0xffffffffULL)));
        }
    } else {
        // ---- ticket hand-off ----
        __shared__ unsigned int s_last;
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < NRED; ++k) store_agent(red.partials + (size_t)k * MAX_GRID + blockIdx.x, acc[k]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // partials have left this CU
            unsigned int t = __hip_atomic_fetch_add(red.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == G - 1u) ? 1u : 0u;
        }
        __syncthreads();
        if (s_last == 0u) return;  // uniform per workgroup
        // last workgroup: every other workgroup's partials were drained before its ticket add
#pragma unroll
        for (int k = 0; k < NRED; ++k) {
            double t = 0.0;
            for (unsigned int b = threadIdx.x; b < G; b += BLOCK) t += load_agent(red.partials + (size_t)k * MAX_GRID + b);
            tot[k] = t;
        }
        __syncthreads();  // lds reuse
        if (threadIdx.x == 0) __hip_atomic_store(red.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    block_sum<NRED>(tot, lds);
    if (red.p2p.world > 1) {  // close the reduction across ranks before the kernel ends (uniform branch)
        __shared__ double s_vals[NRED];
        __shared__ unsigned int s_bits[P2P_MAX_WORLD][MAX_RED][2];
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < NRED; ++k) s_vals[k] = tot[k];
        }
        __syncthreads();
        // (DevXchg: the record is loaded before the exchange -- its latency hides behind the peers' -- and stored after it)
        DevXchg* const xs = dev_xchg(red.ctr, red.xchg_class);
        DevXchg x{};
        long long t1 = 0;
        if (threadIdx.x == 0) {
            x = *xs;
            t1 = wall_clock64();
        }
        p2p_exchange(red.p2p, c0.p2p_epoch, s_vals, NRED, s_bits);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < NRED; ++k) tot[k] = s_vals[k];
            red.ctr->p2p_epoch = next_epoch(c0.p2p_epoch);
            const unsigned long long dt_x = (unsigned long long)(wall_clock64() - t1), dt_l = tagged ? (unsigned long long)(t1 - t0) : 0ull;
            x.p2p_ticks += dt_x;
            x.local_ticks += dt_l;
            x.count += 1ull;
            *xs = x;
            xchg_dist_note(red.ctr, red.xchg_class, dt_x, dt_l);
        }
    }
    if (threadIdx.x == 0) {
        if (tagged) red.ctr->red_epoch = next_epoch(epoch);
#pragma unroll
        for (int k = 0; k < NRED; ++k) {
            if constexpr (NRED <= RED_PTRS) *red.out[k] = tot[k];
            else red.out_contig[k] = tot[k];
        }
        if (red.dup_ptr) *red.dup_ptr = tot[red.dup_k < NRED ? red.dup_k : 0];
        if constexpr (NRED <= RED_PTRS) {
            if (red.mirror.host_board) {  // publish the final totals to the host (system scope), then the sequence
#pragma unroll
                for (int k = 0; k < NRED; ++k) {
                    const long idx = red.out[k] - red.mirror.board;
                    if (idx >= 0 && idx < red.mirror.slots)
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(red.mirror.host_board + idx),
                                           (unsigned long long)__double_as_longlong(tot[k]), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (red.mirror.host_err) {  // a timed-out exchange or hand-off travels with the results
                    const unsigned int e = __hip_atomic_load(red.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (e)
                        __hip_atomic_store(red.mirror.host_err, (unsigned long long)e, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                }
                // Order "results, then sequence word" WITHOUT a release fence: a system-scope release makes the compiler
                // write back this XCD's whole L2 (buffer_wbl2 sc0 sc1) -- megabytes of the running vector's dirty lines,
                // microseconds on every reducing kernel -- although nothing the host reads lives in L2: the stores above
                // are system-scope atomics (written through).  Waiting for them to complete (vmcnt) before issuing the
                // sequence store gives the same order at the host.
                const unsigned long long seq = c0.mirror_seq + 1ull;
                red.ctr->mirror_seq = seq;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(red.mirror.host_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// 16-byte global access; NT adds the `nt` (streaming, no reuse expected) cache hint
template <bool NT>
__device__ __forceinline__ d2 ld16(const double* base, uint64_t pair) {
    const d2* p = reinterpret_cast<const d2*>(base) + pair;
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
#ifndef LH_STORE_POLICY
#define LH_STORE_POLICY 0   // tuning experiments only (tools/tune_stream.hip): 1 = sc1, 2 = sc0 sc1, 3 = nt sc1
#endif
template <bool NT>
__device__ __forceinline__ void st16(double* base, uint64_t pair, d2 v) {
    d2* p = reinterpret_cast<d2*>(base) + pair;
#if LH_STORE_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif LH_STORE_POLICY == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif LH_STORE_POLICY == 3
    asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
#else
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
#endif
}

// per-stream hint selection (s is a compile-time constant after unrolling)
template <unsigned MASK>
__device__ __forceinline__ d2 ld_masked(const double* base, uint64_t pair, int s) {
    if constexpr (MASK == 0u) return ld16<false>(base, pair);
    else if constexpr (MASK == ~0u) return ld16<true>(base, pair);
    else return ((MASK >> s) & 1u) ? ld16<true>(base, pair) : ld16<false>(base, pair);
}
template <unsigned MASK>
__device__ __forceinline__ void st_masked(double* base, uint64_t pair, d2 v, int s) {
    if constexpr (MASK == 0u) st16<false>(base, pair, v);
    else if constexpr (MASK == ~0u) st16<true>(base, pair, v);
    else if ((MASK >> s) & 1u) st16<true>(base, pair, v);
    else st16<false>(base, pair, v);
}

// ---- the skeleton ----------------------------------------------------------------------
// Op interface:
//   static constexpr int NIN, NOUT, NRED;
//   const double* in[NIN]; double* out[NOUT];
//   typename Op::Coef; __device__ Coef setup() const;            // wave-uniform scalars (board reads)
//   __device__ void elem(const Coef&, const double* v, double* w, double* acc, uint64_t gidx) const;
//     v[NIN] inputs of one element, w[NOUT] outputs, acc[NRED] running sums, gidx = GLOBAL index
// An Op with `static constexpr bool PAIRWISE = true` instead implements
//   __device__ void pair(const Coef&, const d2* v, d2* w, double* acc, uint64_t gidx0) const;
template <class Op>
struct is_pairwise {
    template <class T> static constexpr bool test(decltype(T::PAIRWISE)*) { return T::PAIRWISE; }
    template <class T> static constexpr bool test(...) { return false; }
    static constexpr bool value = test<Op>(nullptr);
};

// An Op may pin its own (MAP, UNROLL) with `static constexpr int TUNE_MAP, TUNE_UNROLL` (measured per kernel
// shape with tools/tune_stream.hip); otherwise the defaults apply.
template <class Op>
struct tuning {
    template <class T> static constexpr int map(decltype(T::TUNE_MAP)*) { return T::TUNE_MAP; }
    template <class T> static constexpr int map(...) { return DEFAULT_MAP; }
    template <class T> static constexpr int unr(decltype(T::TUNE_UNROLL)*) { return T::TUNE_UNROLL; }
    template <class T> static constexpr int unr(...) { return UNROLL; }
    // workgroups per 32 CUs (TUNE_GRID_X32): 27 for the bandwidth-bound shapes (more workgroups only add concurrent
    // DRAM pages); operators with real arithmetic per element (hashed objectives) want >= 2 workgroups per CU
    template <class T> static constexpr int gx32(decltype(T::TUNE_GRID_X32)*) { return T::TUNE_GRID_X32; }
    template <class T> static constexpr int gx32(...) { return 27; }
    static constexpr int MAP = map<Op>(nullptr);
    static constexpr int UNR = unr<Op>(nullptr);
    static constexpr int GRID_X32 = gx32<Op>(nullptr);
};

// Operators that hash the GLOBAL index (the synthetic objectives of ops.h: data from splitmix64 of the index, nothing stored)
// declare `static constexpr bool INDEX_PRODUCT = true` and take, next to the index, its product (gidx + 1) * IDX_GOLDEN mod 2^64
// -- the first step of that hash -- from the skeleton: `elem(cf, v, w, acc, gidx, hp)`.  A 64-bit multiply is three
// quarter-rate integer multiplies on gfx950; the skeleton knows that the index products of a trip differ by constants, forms
// ONE per thread and trip (8 elements) and adds (bit-identical: arithmetic mod 2^64).  Worth 8 % on the trial probe of the
// hashed quadratic, whose ALU work sits close behind its two reads.
constexpr uint64_t IDX_GOLDEN = 0x9E3779B97F4A7C15ULL;
template <class Op>
struct wants_index_product {
    template <class T> static constexpr bool test(decltype(T::INDEX_PRODUCT)*) { return T::INDEX_PRODUCT; }
    template <class T> static constexpr bool test(...) { return false; }
    static constexpr bool value = test<Op>(nullptr);
};

// hp0: (gidx0 + 1) * IDX_GOLDEN for operators that want it (anything otherwise)
template <class Op>
__device__ __forceinline__ void do_pair(const Op& op, const typename Op::Coef& cf, const d2* v, d2* w, double* acc,
                                        uint64_t gidx0, uint64_t hp0) {
    if constexpr (is_pairwise<Op>::value) {
        op.pair(cf, v, w, acc, gidx0);
    } else if constexpr (wants_index_product<Op>::value) {
        double a[Op::NIN ? Op::NIN : 1], b[Op::NIN ? Op::NIN : 1];
        double wa[Op::NOUT ? Op::NOUT : 1], wb[Op::NOUT ? Op::NOUT : 1];
#pragma unroll
        for (int s = 0; s < Op::NIN; ++s) { a[s] = v[s].x; b[s] = v[s].y; }
        op.elem(cf, a, wa, acc, gidx0, hp0);
        op.elem(cf, b, wb, acc, gidx0 + 1, hp0 + IDX_GOLDEN);
#pragma unroll
        for (int s = 0; s < Op::NOUT; ++s) { w[s].x = wa[s]; w[s].y = wb[s]; }
    } else {
        double a[Op::NIN ? Op::NIN : 1], b[Op::NIN ? Op::NIN : 1];
        double wa[Op::NOUT ? Op::NOUT : 1], wb[Op::NOUT ? Op::NOUT : 1];
#pragma unroll
        for (int s = 0; s < Op::NIN; ++s) { a[s] = v[s].x; b[s] = v[s].y; }
        op.elem(cf, a, wa, acc, gidx0);
        op.elem(cf, b, wb, acc, gidx0 + 1);
#pragma unroll
        for (int s = 0; s < Op::NOUT; ++s) { w[s].x = wa[s]; w[s].y = wb[s]; }
    }
}

// MAP selects the workgroup -> address map:
//   0  each workgroup owns ONE contiguous, evenly sized range (G*streams concurrent DRAM streams);
//   1  grid-stride over 4 KiB chunks: at any time the whole grid sweeps one contiguous window
//      of UNROLL*G chunks per vector (few, long DRAM streams);
//   2  grid-stride over SPAN*UNROLL*4 KiB super-chunks (each workgroup reads SPAN*UNROLL adjacent
//      chunks, UNROLL at a time).
// NTI / NTO are bit masks over the input / output streams: bit s set = stream s uses the `nt` hint.
// The sweep itself: this workgroup's share of the n elements, running sums into acc[].  Shared by the one-launch-
// per-operator kernel below and by experiments that call it from other kernels (tools/tune_stream.hip).
// `coef()` yields the operator's wave-uniform coefficients (Op::setup(): board reads + a division).  It is called AFTER
// the first trip's loads have been issued, so that the ~1 us it takes (dependent scalar loads of values the previous
// kernel has just written) overlaps the first vector loads instead of preceding them.
// LATE = false keeps the plain order (coefficients, then the sweep): on vectors of hundreds of MB the overlap buys
// nothing and the longer-lived first-trip registers cost ~0.5 % (profiles/r02_late_setup_ab.log).
template <class Op, int UNR, unsigned NTI, unsigned NTO, int MAP, int SPAN, bool LATE, class CoefFn>
__device__ __forceinline__ void stream_sweep(const Op& op, CoefFn coef, const uint64_t n, const uint64_t gofs, double* acc) {
    constexpr int UNROLL = UNR;
    constexpr int NIN = Op::NIN, NOUT = Op::NOUT;
    const uint64_t n2 = n >> 1;                           // 16-byte pairs
    const uint64_t nch = (n2 + BLOCK - 1) / BLOCK;        // chunks of BLOCK pairs (4 KiB per stream)
    const uint64_t tid = threadIdx.x;
    const uint64_t G = gridDim.x, B = blockIdx.x;

    // chunk index of this workgroup's u-th chunk in trip t, and the trip count
    uint64_t c0 = 0, c1 = 0, trips = 0;
    if constexpr (MAP == 0) {
        c0 = nch * B / G;                                 // contiguous chunk range [c0, c1)
        c1 = nch * (B + 1) / G;
        trips = (c1 - c0 + UNROLL - 1) / UNROLL;
    } else {
        const uint64_t per_round = G * UNROLL * (MAP == 2 ? SPAN : 1);
        trips = (nch + per_round - 1) / per_round * (MAP == 2 ? SPAN : 1);
    }
    auto chunk_of = [&](uint64_t t, int u) -> uint64_t {
        if constexpr (MAP == 0) return c0 + t * UNROLL + u;
        else if constexpr (MAP == 1) return (t * UNROLL + u) * G + B;
        else return (((t / SPAN) * G + B) * SPAN + (t % SPAN)) * UNROLL + u;
    };
    auto chunk_end = [&]() -> uint64_t { return MAP == 0 ? c1 : nch; };

    auto full_trip = [&](uint64_t t) -> bool {  // UNROLL whole chunks: no bounds checks needed
        const uint64_t last = chunk_of(t, UNROLL - 1);
        return last < chunk_end() && (last + 1) * BLOCK <= n2;
    };
    auto load_trip = [&](uint64_t t, d2 (&v)[UNROLL][NIN ? NIN : 1]) {  // all loads of a trip, issued back to back
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t p = chunk_of(t, u) * BLOCK + tid;
#pragma unroll
            for (int s = 0; s < NIN; ++s) v[u][s] = ld_masked<NTI>(op.in[s], p, s);
        }
    };
    auto work_trip = [&](const typename Op::Coef& cf, uint64_t t, d2 (&v)[UNROLL][NIN ? NIN : 1]) {
        // (index products: the chunks of a trip are a fixed stride apart -- 1, or G under MAP 1 -- so one multiply serves the trip)
        uint64_t hp_trip = 0;
        if constexpr (wants_index_product<Op>::value) hp_trip = (gofs + 2 * (chunk_of(t, 0) * BLOCK + tid) + 1) * IDX_GOLDEN;
        const uint64_t hp_chunk = (MAP == 1 ? G : 1) * (2ull * BLOCK) * IDX_GOLDEN;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t p = chunk_of(t, u) * BLOCK + tid;
            d2 w[NOUT ? NOUT : 1];
            do_pair<Op>(op, cf, v[u], w, acc, gofs + 2 * p, hp_trip + (uint64_t)u * hp_chunk);
#pragma unroll
            for (int s = 0; s < NOUT; ++s) st_masked<NTO>(op.out[s], p, w[s], s);
        }
    };
    auto ragged_trip = [&](const typename Op::Coef& cf, uint64_t t) {  // chunk by chunk, pair by pair
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t c = chunk_of(t, u);
            const uint64_t p = c * BLOCK + tid;
            if (c < chunk_end() && p < n2) {
                d2 v[NIN ? NIN : 1], w[NOUT ? NOUT : 1];
#pragma unroll
                for (int s = 0; s < NIN; ++s) v[s] = ld_masked<NTI>(op.in[s], p, s);
                do_pair<Op>(op, cf, v, w, acc, gofs + 2 * p, (gofs + 2 * p + 1) * IDX_GOLDEN);
#pragma unroll
                for (int s = 0; s < NOUT; ++s) st_masked<NTO>(op.out[s], p, w[s], s);
            }
        }
    };
    // first trip: loads first, coefficients second
    d2 v0[UNROLL][NIN ? NIN : 1];
    const bool first_full = LATE && trips > 0 && full_trip(0);
    if (first_full) load_trip(0, v0);
    const typename Op::Coef cf = coef();
    if (first_full) work_trip(cf, 0, v0);
    else if (LATE && trips > 0) ragged_trip(cf, 0);
    for (uint64_t t = LATE ? 1 : 0; t < trips; ++t) {
        if (full_trip(t)) {
            // fast path: UNROLL full chunks, all loads issued before the first use
            d2 v[UNROLL][NIN ? NIN : 1];
            load_trip(t, v);
            work_trip(cf, t, v);
        } else {
            ragged_trip(cf, t);
        }
    }
    // odd n: the last element, scalar (never taken by PAIRWISE ops: their n is even)
    if constexpr (!is_pairwise<Op>::value) {
        if ((n & 1) && blockIdx.x == gridDim.x - 1 && tid == 0) {
            const uint64_t e = n - 1;
            double a[NIN ? NIN : 1], wa[NOUT ? NOUT : 1];
#pragma unroll
            for (int s = 0; s < NIN; ++s) a[s] = op.in[s][e];
            op.elem(cf, a, wa, acc, gofs + e);
#pragma unroll
            for (int s = 0; s < NOUT; ++s) op.out[s][e] = wa[s];
        }
    }
}

// the sweep with coefficients that are already known (experiments in tools/tune_stream.hip)
template <class Op, int UNR, unsigned NTI, unsigned NTO, int MAP, int SPAN>
__device__ __forceinline__ void stream_body(const Op& op, const typename Op::Coef& cf, const uint64_t n,
                                            const uint64_t gofs, double* acc) {
    stream_sweep<Op, UNR, NTI, NTO, MAP, SPAN, false>(op, [&]() { return cf; }, n, gofs, acc);
}

template <class Op, int UNR = UNROLL, unsigned NTI = 0, unsigned NTO = 0, int MAP = 0, int SPAN = 1, bool LATE = false>
__global__ __launch_bounds__(BLOCK) void stream_kernel(const Op op, const uint64_t n, const uint64_t gofs,
                                                        const RedCtl red) {
    constexpr int NRED = Op::NRED;
    static_assert(NRED <= MAX_RED, "partials buffer overflow");
    DevCounters c0{};
    if constexpr (NRED > 0) c0 = load_counters(red);
    double acc[NRED ? NRED : 1];
#pragma unroll
    for (int k = 0; k < (NRED ? NRED : 1); ++k) acc[k] = 0.0;
    stream_sweep<Op, UNR, NTI, NTO, MAP, SPAN, LATE>(op, [&]() { return op.setup(); }, n, gofs, acc);
    if constexpr (NRED > 0) grid_reduce<NRED>(reinterpret_cast<double(&)[NRED]>(acc), red, c0);
}

}  // namespace lh
