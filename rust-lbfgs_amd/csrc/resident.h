// rust-lbfgs_amd/csrc/resident.h -- the two-loop recursion (lbfgs.rs:569-604) as ONE persistent kernel that keeps the
// running vector q ON THE CHIP for the whole recursion -- all of it when the shard fits, the first part of it otherwise.
//
// MI355X has 128 MiB of vector registers (256 CUs x 4 SIMD x 512 x 64 lanes x 4 B) and 40 MiB of LDS.  The running
// vector of the recursion -- read and rewritten by every one of its 2*bound steps -- fits there whenever a rank's shard
// is at most ~1.25e7 elements (100 MB): each of 256 workgroups x 256 threads holds ER 16-byte pairs in registers and up
// to EL more in LDS.  Then a step streams only its TWO history vectors from HBM (q += c*u ; out = v.q) instead of three
// reads and a write, and the steps are separated by a grid-wide hand-off of the partial sums instead of a kernel
// boundary:
//      launch-per-step path   8*bound - 1 passes of an n-vector, 2*bound kernel boundaries
//      this kernel            4*bound + 1 passes (g at the start, every s and y in both loops, g again in the last step,
//                             d written once), one launch
// That is the per-rank regime of the 8-GPU run of BASELINE.json's metric (n = 1e8 / 8), and of its configs 2 and 3.
// HYBRID: a larger shard keeps the first ER + EL = 96 rounds of every thread on the chip and leaves the rest of q in
// HBM (in `d`), streamed 3r + 1w by every step as on the launch-per-step path -- (4b+1) passes over the on-chip
// elements, (8b-1) over the others, still one launch (n = 1e8 on one GPU: 10.6 % of q on chip; res_step_hbm below).
//
// Synchronisation between steps.  Nothing but the partial sums crosses workgroups: q never leaves its thread (the HBM
// part of a hybrid shard is read and rewritten by the same thread every step), u / v / g were written by earlier
// kernels.  After a step every workgroup publishes its partial sum(s) as tagged 8-byte
// granules (tag = the step's sequence number; agent-scope atomics on both sides, no fence: stream.h) into a double
// buffer indexed by the step's parity, and then EVERY workgroup collects all G partials -- thread t polls workgroup t's
// granules (touching kernels: t's and t + 128's, see TOUCHING) -- and adds them up in one fixed order (each wave's 64 values by stream.h wave_sum_dpp, then the waves in
// order).  All workgroups therefore hold the same bits for the total and form the same coefficient; no broadcast step
// is needed.  (That order is this kernel's own: not the launch-per-step reducer's ds_bpermute tree.)  A workgroup can
// be at most one step ahead of any other (it needs everybody's partial of step s to leave step s), so two buffers
// suffice.  All G workgroups must be resident at once: the grid is at most one workgroup per CU (the LDS share makes it
// exactly one), the kernel is launched alone on its stream, and every spin is bounded by a wall-clock timeout that raises
// *red.err (surfaced at the next scalar read) -- a missing workgroup ends as an error, not as a hang.
//
// Arithmetic is the reference's, operation by operation (q + c*u: a multiply then an add, math.rs:35; dot products as
// running sums, math.rs:41; -ffp-contract=off); only the order of the dot products' partial sums differs from the
// launch-per-step kernels (other element -> thread map), as it differs between any two grids.
#pragma once
#include "stream.h"

namespace lh {

constexpr int RES_MAX_STEPS = 2 * 24;  // 2 * bound for bound <= 24 (the step table travels as a kernel argument: < 4 KiB)
constexpr int RES_LDS_PAIRS_MAX = 36;  // 160 KiB / (256 threads x 16 B) = 40, less the kernel's static arrays (4.3 KiB),
                                       // rounded down to a multiple of RES_UNROLL
constexpr int RES_UNROLL = 4;          // pairs whose loads are issued together (one group)
// TOUCHING.  While a workgroup waits in a hand-off, HBM idles: the NEXT step's operands are known (only its coefficient is
// not), but the registers and LDS are full of q, so only a window of RES_AHEAD groups can be loaded ahead.  So a waiting
// workgroup TOUCHES the lines of the rounds that follow that window -- one 4-byte load per 128-byte line, result unused --
// and finds them in its XCD's L2 when the step asks for them.  A wave's loads return in order, so a touch must never sit in
// front of a poll whose answer is on the critical path: the workgroup's waves SPLIT THE ROLES.  The lower two poll (thread t
// the granules of workgroups t and t + 128) and never touch; the upper two touch and never poll, and go straight to the
// barrier in front of the workgroup's sum with their touches in flight.  (The form in which every thread polled one
// workgroup and issued touches behind its first poll made every later poll wait for HBM: 0.6 us more until the last
// workgroup had left a hand-off; n = 1.25e7: 618.8 -> 607.4 us, 1.2e6: 69.3 -> 66.2 us, profiles/r04_split_ab.log.)
// Depth (ResArgs::touch_rounds, chosen by the host): 8 rounds up to ~60 rounds per thread, 16 beyond -- 16 rounds of two
// vectors are 4 MiB per XCD, its whole L2; 24 and more thrash (n = 1.25e7: 662 -> 622 us with 16, 679 with 24, 810 with 48;
// n = 3e6: 129 -> 119 us with 8, 128 with 16; profiles/r04_touch_sweep.log).  Compiled in for every kernel whose shards have
// something behind the window (ER >= 8, not hybrid: hybrid steps take hundreds of microseconds, nothing to hide there).
// LH_RES_TOUCH=0 builds the kernels without it (A/B).
#ifndef LH_RES_TOUCH
#define LH_RES_TOUCH 16  // the deepest touch compiled in (rounds, a multiple of 4): a touching thread holds LH_RES_TOUCH / 2 words
#endif
#ifndef LH_RES_D_NT
#define LH_RES_D_NT 1  // the direction's write-out pass uses `nt` stores where the history loads use `nt`
#endif
#ifndef LH_RES_TRACE
#define LH_RES_TRACE 0  // 1: every workgroup logs wall-clock stamps of every hand-off into RedCtl::partials (tools/handoff_trace.sh)
#endif
// What was measured and dropped (profiles/EXPERIMENTS.md has the numbers; the code is gone): two groups in flight ahead of the
// one worked on (RES_AHEAD = 2: no faster at 1.25e7 elements, 4 % slower at 3e6); the next step's window loaded after the
// hand-off, or inside the step's last on-chip group, instead of right before the hand-off; the round-2 hand-off with five
// barriers and ds_bpermute wave sums; a one-hop P2P form in which every workgroup reads the mailbox; the plain last step
// storing q as it completes it (1.5-3.6 % slower than the separate pass of stores); HBM rounds swept upwards in every step.
constexpr int RES_HBM_UNROLL = 4;  // hybrid: rounds per group of the part of q that stays in HBM
constexpr int RES_AHEAD = 1;       // groups whose loads are in flight ahead of the group being worked on

struct ResStep {
    const double* u;     // the vector added to q:   q += c * u
    const double* v;     // the vector the new q is multiplied with (nullptr: with u itself -- the gamma transition)
    const double* ys;    // ys of the correction whose coefficient this step applies (lbfgs.rs:587,597)
    double* alpha;       // its alpha: written in mode A (first loop), for API parity with the other paths
    int mode_b;          // 0: c = -(dot/ys), alpha = dot/ys     1: c = alpha - dot/ys
    int scale;           // 1: q *= gamma after the update (lbfgs.rs:591)
    int alpha_idx;       // position in this kernel's private alpha stack
    int last;            // 1: the last step -- sums ||q||^2 and v.q (v = g: the next line search's dginit)
};

struct ResArgs {
    const double* g;         // q starts as -g (core.rs:95-101)
    double* d;               // the result
    const double* first_dot; // s_{j0}.(-g) if somebody has summed it already (the history-update kernel), else nullptr
    const double* first_s;   // ... else it is summed here: s_{j0}
    const double* gnum;      // gamma = *gnum / *gden (lbfgs.rs:691)
    const double* gden;
    double* out_dn;          // ||d||^2 ; out_dn[1] = g.d
    uint64_t n;              // elements of this rank's shard
    uint64_t gofs;           // global index of this shard's first element (OWL-QN's range is global)
    uint64_t owl_start, owl_end;  // OWL-QN: the last step also projects d onto the orthant of -pg on [start, end)
    int owl;                      //   (orthantwise.rs:140-161; then v of the last step is pg) and leaves 3 sums
    uint32_t pairs_per_thread;  // E: 16-byte pairs each thread owns (registers first, then LDS, then -- hybrid -- HBM)
    uint32_t lds_pairs;         // of which in LDS (a multiple of RES_UNROLL)
    uint32_t hbm_pairs;         // hybrid: the rounds beyond registers + LDS, whose part of q lives in `d` itself (0: none)
    uint32_t hbm_plain_pairs;   // ... of which the first so many are accessed with the default cache policy (res_hbm_rounds)
    uint32_t touch_rounds;      // rounds behind the window that a waiting workgroup touches (TOUCHING above; <= LH_RES_TOUCH)
    int nsteps;
    int fault_drop_last;        // tests (LBFGS_HIP_RESIDENT_FAULT=1): the last workgroup leaves at once, as if it had never been
                                // given a CU -- the others time out in their first hand-off and the host falls back
    ResStep step[RES_MAX_STEPS];
};

// what a waiting workgroup touches (LH_RES_TOUCH): rounds [r0, r_end) of the next step's u and v, this workgroup's 4 KiB chunk of each
struct ResTouch {
    const char* u = nullptr;  // (always valid vectors: after the last step, vectors whose lines are in the caches anyway)
    const char* v = nullptr;
    uint32_t first = 0, round_stride = 0;  // byte offset of this workgroup's chunk in round 0; bytes from one round to the next
    uint32_t r0 = 0, r_end = 0;
    uint32_t limit = 0;                    // the last byte offset that may be read
};
// The touched words land in `sink` and stay there, unread, until the step that follows has consumed operands it loaded LATER
// (res_touch_retire: loads return in order, so the touches have returned by then and nothing ever waits for them).  Plain
// loads on purpose: a volatile load is compiled to a system-scope access followed by s_waitcnt vmcnt(0).
constexpr int RES_TOUCHERS = BLOCK / 2;  // threads of a workgroup that touch: its upper two waves (the lower two poll; res_exchange)
constexpr int RES_TOUCH_REGS = LH_RES_TOUCH > 0 ? (2 * 32 * LH_RES_TOUCH + RES_TOUCHERS - 1) / RES_TOUCHERS : 1;
struct ResSink {
    unsigned int w[RES_TOUCH_REGS];
};
// `tt`: this thread's index among the touching threads (0 .. RES_TOUCHERS-1)
template <int D>
__device__ __forceinline__ void res_touch(const ResTouch& t, ResSink& sink, const uint32_t tt) {
    if constexpr (D > 0) {
        // (whole multiples of the touching threads per vector: which vector a thread touches is then a compile-time fact -- a
        // run-time choice between the two pointers is compiled to an indexed read of this struct, i.e. to scratch memory)
        static_assert((32 * D) % RES_TOUCHERS == 0, "LH_RES_TOUCH must be a multiple of the touching threads / 32");
        constexpr int PER_VEC = 32 * D / RES_TOUCHERS;
        // STRAIGHT-LINE code on purpose: a branch around a trip (tried in round 5 to skip trips beyond a run-time depth) ends in a
        // join, at which the compiler waits for every load in flight -- the touches land on the critical path and the kernel is
        // 1-2 % slower (profiles/r05_build_ab.log).  The depth is a compile-time fact of the instantiation instead (D).
#pragma unroll
        for (int i = 0; i < 2 * PER_VEC; ++i) {
            const uint32_t L = tt + (uint32_t)(RES_TOUCHERS * (i % PER_VEC));  // line among this vector's 32 * D
            const uint32_t r = min(t.r0 + L / 32u, t.r_end - 1u), line = L % 32u;
            const uint32_t off = min(t.first + r * t.round_stride + line * 128u, t.limit);
            sink.w[i] = *reinterpret_cast<const unsigned int*>((i < PER_VEC ? t.u : t.v) + off);
        }
    }
}
__device__ __forceinline__ void res_touch_retire(const ResSink& sink) {
    if constexpr (LH_RES_TOUCH > 0) {
#pragma unroll
        for (int i = 0; i < RES_TOUCH_REGS; ++i) asm volatile("" ::"v"(sink.w[i]));
    }
}

// Publish this workgroup's NS partial sums of sequence number `tag`, then collect everybody's and return the totals
// (identical bits in every workgroup).  gran rows: [parity*4 + k], k < 4.  All threads call it.
//   one rank      EVERY workgroup polls all G partials itself and adds them up: one hop.
//   several ranks only workgroup 0 does that; it then exchanges this rank's totals with the peers through the xGMI
//                 mailboxes (stream.h p2p_exchange, epoch `p2p_tag`; rank-ordered sum: the same bits on every rank) and
//                 publishes the global totals as tagged granules in rows [8 + parity*4 + k], which the other workgroups
//                 poll.  The mailbox is uncached and system-scope: one poller per GPU, not 256.
// The hand-off is pure latency (nothing streams while a workgroup waits), so it is built from as few serial pieces as
// possible: the workgroup's sums are formed with ONE barrier each (every thread adds the four wave sums itself, in
// block_sum's order -- no second trip through LDS to broadcast), the wave sums use DPP moves instead of ds_bpermute trees
// (stream.h wave_sum_dpp), and the two sums use different LDS rows, so no barrier separates one hand-off from the next:
// three barriers per hand-off (round 2: five).  The granules live in uncached memory (lbfgs_hip_ctx_create).
// `lds`: [8][WAVES], rows 0-3 for the sums a workgroup publishes, rows 4-7 for the totals it collects.  A row is written
// before one of the two barriers and read right after it; its next write lies behind the OTHER barrier of the same or the
// next hand-off, which no thread passes before every thread has finished that read.
template <int NS>
__device__ __forceinline__ void res_block_total(double (&acc)[NS], double (*rows)[WAVES]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double w = wave_sum_dpp(acc[k]);
        if (lane == 0) rows[k][wave] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NS; ++k) {  // block_sum's order: the waves' sums, wave 0 first
        double t = rows[k][0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t += rows[k][w];
        acc[k] = t;
    }
}
#if LH_RES_TRACE
#define LH_TR(...) __VA_ARGS__
#else
#define LH_TR(...)
#endif
// what workgroup 0 of a rank spent in the hand-offs of one launch (several ranks only): DevXchg's figures, kept in registers
// until the kernel ends
struct ResXchgAcc {
    unsigned long long p2p_ticks = 0, local_ticks = 0, count = 0;
};
template <int NS, int TOUCH>
__device__ __forceinline__ void res_exchange(double (&acc)[NS], const RedCtl& red, const unsigned int tag, const unsigned int p2p_tag,
                                             const int parity, double (*lds)[WAVES], double* s_tot,
                                             unsigned int (*s_bits)[MAX_RED][2], ResSink& sink, ResXchgAcc& xacc,
                                             const ResTouch& touch) {
    LH_TR(const long long tr0 = wall_clock64(); long long tr1 = 0; unsigned int tr_polls = 0;)
    res_block_total<NS>(acc, lds);  // (every thread holds the workgroup's sums)
    const unsigned int G = gridDim.x;  // (<= BLOCK: the host launches no larger grid -- thread t polls workgroup t)
    const bool multi = p2p_active(red.p2p);
    const unsigned long long t = (unsigned long long)tag << 32;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(acc[k]);
            unsigned long long* g = red.gran + ((size_t)(parity * 4 + k) * MAX_GRID + blockIdx.x) * 2;
            __hip_atomic_store(g, t | (b & 0xffffffffULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 1, t | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    LH_TR(asm volatile("" ::: "memory"); tr1 = wall_clock64();)
    // Nobody polls before its own workgroup's partial is on its way.  (Without this barrier 255 threads of every workgroup
    // start polling while thread 0 is still publishing, and 65 000 early pollers on the 32 cache lines that hold the
    // granules slow every store to them down: 2.5 us per hand-off against 1.7, tools/handoff_bench.hip.)
    __syncthreads();
    double (*rows2)[WAVES] = lds + 4;
    double tot[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) tot[k] = 0.0;
    const long long t0 = wall_clock64();
    const bool collector = !multi || blockIdx.x == 0;  // (uniform) this workgroup adds up the G partials itself
    if (collector && TOUCH > 0) {
        // SPLIT ROLES: the lower half of the waves polls -- thread t the granules of workgroups t and t + BLOCK/2 --, the upper
        // half touches and does not poll.  A wave's loads return in order: a thread that touched would read its LATER polls only
        // after the touches have come back from HBM (the all-waves form: +0.6 us until the last workgroup has left a
        // hand-off); a polling wave with nothing else in its queue reads every poll as soon as the memory answers.
        if (threadIdx.x < (unsigned int)(BLOCK / 2)) {
            const unsigned int b1 = threadIdx.x, b2 = threadIdx.x + (unsigned int)(BLOCK / 2);
            const bool mine1 = b1 < G, mine2 = b2 < G;
            const unsigned int bb[2] = {mine1 ? b1 : blockIdx.x, mine2 ? b2 : blockIdx.x};
            unsigned long long lo[2][NS], hi[2][NS];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        const unsigned long long* g = red.gran + ((size_t)(parity * 4 + k) * MAX_GRID + bb[j]) * 2;
                        lo[j][k] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hi[j][k] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < NS; ++k)
                        ok = ok && (unsigned int)(lo[j][k] >> 32) == tag && (unsigned int)(hi[j][k] >> 32) == tag;
                LH_TR(tr_polls++;)
                if (ok) break;
                if ((unsigned long long)(wall_clock64() - t0) > red.timeout_ticks ||
                    __hip_atomic_load(red.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    atomicExch(red.err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const double v1 = __longlong_as_double((long long)((hi[0][k] << 32) | (lo[0][k] & 0xffffffffULL)));
                const double v2 = __longlong_as_double((long long)((hi[1][k] << 32) | (lo[1][k] & 0xffffffffULL)));
                tot[k] = mine1 ? 0.0 + v1 : 0.0;
                if (mine2) tot[k] += v2;
            }
        } else {
            res_touch<TOUCH>(touch, sink, threadIdx.x - (unsigned int)(BLOCK / 2));
        }
    } else if (collector) {
        // (kernels that do not touch: the hybrid ones and those whose shard fits the registers) thread b polls workgroup b's granules
        const bool mine = threadIdx.x < G;
        unsigned long long lo[NS], hi[NS];
        if (mine) {
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const unsigned long long* g = red.gran + ((size_t)(parity * 4 + k) * MAX_GRID + threadIdx.x) * 2;
                    lo[k] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hi[k] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < NS; ++k) ok = ok && (unsigned int)(lo[k] >> 32) == tag && (unsigned int)(hi[k] >> 32) == tag;
                LH_TR(tr_polls++;)
                if (ok) break;
                // (slow path only) give up after the timeout -- or at once if somebody already has: one missing workgroup must
                // cost ONE timeout, not one per hand-off and workgroup
                if ((unsigned long long)(wall_clock64() - t0) > red.timeout_ticks ||
                    __hip_atomic_load(red.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    atomicExch(red.err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k)  // (0.0 + v: the bits of the strided sum this replaces, -0.0 included)
                tot[k] = 0.0 + __longlong_as_double((long long)((hi[k] << 32) | (lo[k] & 0xffffffffULL)));
        }
    }
    if (collector) {
        // a fixed order: the partials of a thread, each wave's sum (stream.h wave_sum_dpp: rows of 16 as trees, the four rows
        // in order), the waves in order -- the same in every workgroup, so all of them hold the same bits
        res_block_total<NS>(tot, rows2);
        if (multi) {  // (workgroup 0 only) this rank's totals -> the global totals, then tell the other workgroups
            if (threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < NS; ++k) s_tot[k] = tot[k];
            }
            __syncthreads();
            const long long t1 = wall_clock64();
            if (red.p2p.ext_buf) ext_exchange(red.p2p, p2p_tag, s_tot, NS);  // (gated: ncclAllReduce on the second stream)
            else p2p_exchange(red.p2p, p2p_tag, s_tot, NS, s_bits);
            const unsigned long long dt_x = (unsigned long long)(wall_clock64() - t1), dt_l = (unsigned long long)(t1 - t0);
            xacc.p2p_ticks += dt_x;
            xacc.local_ticks += dt_l;
            xacc.count += 1ull;
            if (threadIdx.x == 0) xchg_dist_note(red.ctr, 1u, dt_x, dt_l);  // (fire-and-forget atomics: the distribution behind the totals)
            if (threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const unsigned long long b2 = (unsigned long long)__double_as_longlong(s_tot[k]);
                    unsigned long long* g = red.gran + ((size_t)(8 + parity * 4 + k) * MAX_GRID) * 2;
                    __hip_atomic_store(g, t | (b2 & 0xffffffffULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(g + 1, t | (b2 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    } else {
        // (several ranks, not workgroup 0: the lanes that poll the global total below sit in the first wave, which does not touch
        // in the split form)
        if (threadIdx.x >= (unsigned int)(BLOCK / 2)) res_touch<TOUCH>(touch, sink, threadIdx.x - (unsigned int)(BLOCK / 2));
        if (threadIdx.x < NS) {  // one lane per sum polls the global total workgroup 0 will publish
            const unsigned long long* g = red.gran + ((size_t)(8 + parity * 4 + threadIdx.x) * MAX_GRID) * 2;
            unsigned long long lo, hi;
            for (;;) {
                lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned int)(lo >> 32) == tag && (unsigned int)(hi >> 32) == tag) break;
                if ((unsigned long long)(wall_clock64() - t0) > red.timeout_ticks + red.p2p.timeout_ticks ||
                    __hip_atomic_load(red.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    atomicExch(red.err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            s_tot[threadIdx.x] = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffULL)));
        }
        __syncthreads();
    }
    // (s_tot is written behind a barrier of this hand-off and next written behind one of the next: no barrier needed here)
    if (multi) {
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = s_tot[k];
    } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = tot[k];
    }
#if LH_RES_TRACE
    if (threadIdx.x == 0) {  // every workgroup: [slot = tag & 63][workgroup][4] = arrival, partial published, totals known, polls
        unsigned long long* tr = reinterpret_cast<unsigned long long*>(red.partials) + ((size_t)(tag & 63u) * 256u + blockIdx.x) * 4;
        tr[0] = (unsigned long long)tr0;
        tr[1] = (unsigned long long)tr1;
        tr[2] = (unsigned long long)wall_clock64();
        // polls in the low half; where this workgroup runs in the high half: XCC_ID[3:0] << 16 | HW_ID[15:0] (se, sh, cu, pipe, simd, wave)
        const unsigned int hw = ((unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu) << 16 |
                                ((unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu);
        tr[3] = (unsigned long long)tr_polls | ((unsigned long long)hw << 32);
    }
#endif
}
#undef LH_TR

// One step over this thread's pairs.  MODE 0: q += c*u ; acc0 += v.q     (3 streams -> 2: u and v)
//                                     MODE 1: q = (q + c*u)*gamma ; acc0 += u.q   (the gamma transition)
//                                     MODE 2: q += c*u ; acc0 += q.q ; acc1 += v.q   (the last step, v = g)
// Register rounds are FULL by construction (the host picks ER <= E-1), so they carry no bounds checks at all.
// Under OWL-QN the last step is MODE 2 with v = pg (acc0 = ||d||^2 BEFORE the projection: lbfgs.rs:543 precedes :554); the
// projection d_i = 0 where signum(d_i) != signum(-pg_i) on [start, end) (orthantwise.rs:140-161) is applied where d is
// written out, which sums ||d||^2 and pg.d of the projected direction (pg read once more); all of these sums leave in
// the kernel's last hand-off, four values together.
struct ResOwl {
    uint64_t gofs, start, end;
};
// orthantwise.rs:174-180 as an integer, branch-free: NaN and +-0 -> 0, else the sign
__device__ __forceinline__ int res_sgn(const double x) { return (int)(x > 0.0) - (int)(x < 0.0); }
// d_i = 0 where signum(d_i) != signum(-pg_i), for i in [start, end) (orthantwise.rs:140-161); selects only
__device__ __forceinline__ double res_project(const double q, const double pg, const uint64_t gi, const ResOwl& ow) {
    const bool out = (gi >= ow.start) & (gi < ow.end) & (res_sgn(q) != -res_sgn(pg));
    return out ? 0.0 : q;
}
template <int MODE>
__device__ __forceinline__ void res_one(d2& q, const d2 uu, const d2 vv, const double c, const double gamma, double* acc) {
    q.x = q.x + c * uu.x;  // math.rs:35
    q.y = q.y + c * uu.y;
    if constexpr (MODE == 1) { q.x = q.x * gamma; q.y = q.y * gamma; }  // math.rs:47
    if constexpr (MODE == 2) {
        acc[0] += q.x * q.x; acc[1] += vv.x * q.x;
        acc[0] += q.y * q.y; acc[1] += vv.y * q.y;
    } else if constexpr (MODE == 1) {
        acc[0] += uu.x * q.x; acc[0] += uu.y * q.y;
    } else {
        acc[0] += vv.x * q.x; acc[0] += vv.y * q.y;
    }
}

// ---- the register-resident part of q lives in the ACCUMULATION registers ------------------------------------------
// gfx950 gives a wave that runs alone on its SIMD 512 registers per lane: 256 architectural VGPRs and 256 AGPRs.  The
// compiler allocates the former; the latter it only uses for MFMA and as spill space.  Here pair k of a thread is parked
// in a[4k .. 4k+3] by hand (v_accvgpr_write / _read through inline asm with the register NUMBER as a template constant),
// which leaves all 256 VGPRs to the loads in flight and the arithmetic: 60 pairs = 240 AGPRs = 960 bytes per lane,
// 60 MiB over the chip.  (Register arrays in C++ were tried first: with more than 24 pairs the allocator spills.)
// The translation unit is built with -amdgpu-spill-vgpr-to-agpr=0 so that the compiler never touches an AGPR itself.
template <int IDX>
__device__ __forceinline__ unsigned int acc_rd() {  // (the number is spelled digit by digit: immediates above 9 print in hex)
    unsigned int v;
    if constexpr (IDX < 10) asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(v) : "n"(IDX));
    else if constexpr (IDX < 100) asm volatile("v_accvgpr_read_b32 %0, a%1%2" : "=v"(v) : "n"(IDX / 10), "n"(IDX % 10));
    else asm volatile("v_accvgpr_read_b32 %0, a%1%2%3" : "=v"(v) : "n"(IDX / 100), "n"(IDX / 10 % 10), "n"(IDX % 10));
    return v;
}
template <int IDX>
__device__ __forceinline__ void acc_wr(unsigned int v) {
    if constexpr (IDX < 10) asm volatile("v_accvgpr_write_b32 a%1, %0" ::"v"(v), "n"(IDX));
    else if constexpr (IDX < 100) asm volatile("v_accvgpr_write_b32 a%1%2, %0" ::"v"(v), "n"(IDX / 10), "n"(IDX % 10));
    else asm volatile("v_accvgpr_write_b32 a%1%2%3, %0" ::"v"(v), "n"(IDX / 100), "n"(IDX / 10 % 10), "n"(IDX % 10));
}
template <int K>
__device__ __forceinline__ d2 acc_get() {
    d2 q;
    q.x = __hiloint2double((int)acc_rd<4 * K + 1>(), (int)acc_rd<4 * K>());
    q.y = __hiloint2double((int)acc_rd<4 * K + 3>(), (int)acc_rd<4 * K + 2>());
    return q;
}
template <int K>
__device__ __forceinline__ void acc_put(const d2 q) {
    acc_wr<4 * K>((unsigned int)__double2loint(q.x));
    acc_wr<4 * K + 1>((unsigned int)__double2hiint(q.x));
    acc_wr<4 * K + 2>((unsigned int)__double2loint(q.y));
    acc_wr<4 * K + 3>((unsigned int)__double2hiint(q.y));
}

// a 16-byte access at uniform base + 32-bit byte offset (one VGPR per address: "saddr + voffset")
template <bool NT>
__device__ __forceinline__ d2 ld16_at(const double* base, const uint32_t byte_off) {
    const d2* p = reinterpret_cast<const d2*>(reinterpret_cast<const char*>(base) + byte_off);
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st16_at(double* base, const uint32_t byte_off, const d2 v) {
    d2* p = reinterpret_cast<d2*>(reinterpret_cast<char*>(base) + byte_off);
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// The per-pair offsets are invariant across the steps, so the optimiser would compute all of them ONCE and keep them
// live for the whole kernel -- one or two registers per pair, as much as the state itself.  Passing the first offset
// through an empty asm makes it opaque: every group recomputes its four offsets (four v_add) right where it needs them.
__device__ __forceinline__ uint32_t res_opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
// The running sums are one serial chain of additions per step; left alone, the scheduler (which believes it has 512
// registers to spend) defers that chain and keeps every group's loaded operands alive for it -- over 256 live registers,
// at which point the register allocator parks values in the very AGPRs that hold q.  Pinning the sums at the end of each
// group (volatile asm statements keep their order) makes a group's arithmetic finish before the next group's begins.
__device__ __forceinline__ void res_pin(double& a) { asm volatile("" : "+v"(a)); }
template <int NS>
__device__ __forceinline__ void res_pin_sums(double* acc) {
#pragma unroll
    for (int k = 0; k < NS; ++k) res_pin(acc[k]);
}

// groups of RES_UNROLL pairs, unrolled by template recursion (the AGPR numbers must be compile-time constants)
template <int G, int NG, bool NT, int MODE>
struct ResGroups {
    // q = -g ; acc0 += s * (-g) when the first numerator is summed here
    template <bool NEED_FIRST>
    static __device__ __forceinline__ void init(const uint32_t p_first, const uint32_t p_stride, const double* gp, const double* sp,
                                                double* acc) {
        if constexpr (G < NG) {
            d2 gg[RES_UNROLL], ss[RES_UNROLL];
            const uint32_t o0 = res_opaque(p_first);
#pragma unroll
            for (int u = 0; u < RES_UNROLL; ++u) {
                const uint32_t o = o0 + (uint32_t)(G * RES_UNROLL + u) * p_stride;
                gg[u] = ld16_at<NT>(gp, o);
                if constexpr (NEED_FIRST) ss[u] = ld16_at<NT>(sp, o);
            }
            asm volatile("" ::: "memory");  // (as in step_with: one group of loads at a time)
#pragma unroll
            for (int u = 0; u < RES_UNROLL; ++u) {
                gg[u].x = -gg[u].x;
                gg[u].y = -gg[u].y;
                if constexpr (NEED_FIRST) {  // OpTwoLoopFirst: acc += s * (-g), element by element
                    acc[0] += ss[u].x * gg[u].x;
                    acc[0] += ss[u].y * gg[u].y;
                }
            }
            acc_put<G * RES_UNROLL + 0>(gg[0]);
            acc_put<G * RES_UNROLL + 1>(gg[1]);
            acc_put<G * RES_UNROLL + 2>(gg[2]);
            acc_put<G * RES_UNROLL + 3>(gg[3]);
            if constexpr (NEED_FIRST) res_pin_sums<1>(acc);
            ResGroups<G + 1, NG, NT, MODE>::template init<NEED_FIRST>(p_first, p_stride, gp, sp, acc);
        }
    }
    // d = q (the plain case; under OWL-QN the last step writes q out itself: ResRegStep WO)
    static __device__ __forceinline__ void store(const uint32_t p_first, const uint32_t p_stride, double* dp) {
        if constexpr (G < NG) {
            const uint32_t o0 = res_opaque(p_first);
            st16_at<NT>(dp, o0 + (uint32_t)(G * RES_UNROLL + 0) * p_stride, acc_get<G * RES_UNROLL + 0>());
            st16_at<NT>(dp, o0 + (uint32_t)(G * RES_UNROLL + 1) * p_stride, acc_get<G * RES_UNROLL + 1>());
            st16_at<NT>(dp, o0 + (uint32_t)(G * RES_UNROLL + 2) * p_stride, acc_get<G * RES_UNROLL + 2>());
            st16_at<NT>(dp, o0 + (uint32_t)(G * RES_UNROLL + 3) * p_stride, acc_get<G * RES_UNROLL + 3>());
            ResGroups<G + 1, NG, NT, MODE>::store(p_first, p_stride, dp);
        }
    }
};

// ---- one step = a sequence of GROUPS of RES_UNROLL pairs: NG = ER/RES_UNROLL register groups (compile-time: the AGPR
// numbers are constants), then EL/RES_UNROLL LDS groups (run-time count; the ragged last round is among them:
// bounds-checked).  The operands (u, v) of the next RES_AHEAD groups are always in flight while a group is worked on:
// a window of RES_AHEAD loaded groups travels through the step -- and ACROSS the hand-off: a step's operands are known
// before its coefficient is, so the window of the next step is filled before the partial sums are exchanged and HBM
// keeps streaming while the workgroups wait for each other.
struct ResWin {
    d2 u[RES_AHEAD][RES_UNROLL], v[RES_AHEAD][RES_UNROLL];
};
struct ResPos {
    uint32_t p_first, p_stride, n2, EL;  // first pair / pair stride of this thread, pairs in the shard, pairs in LDS
};
template <int ER, bool NT, bool NEEDV>
struct ResFetch {
    static constexpr int NG = ER / RES_UNROLL;
    template <int GI>
    static __device__ __forceinline__ void reg(const ResPos& ps, const double* up, const double* vp, d2 (&uu)[RES_UNROLL],
                                               d2 (&vv)[RES_UNROLL]) {
        const uint32_t o0 = res_opaque(ps.p_first * 16u);  // byte offsets (shards of < 2^28 pairs: < 2^32 bytes)
#pragma unroll
        for (int u = 0; u < RES_UNROLL; ++u) {
            const uint32_t o = o0 + (uint32_t)(GI * RES_UNROLL + u) * (ps.p_stride * 16u);
            uu[u] = ld16_at<NT>(up, o);
            if constexpr (NEEDV) vv[u] = ld16_at<NT>(vp, o);
            else vv[u] = uu[u];
        }
    }
    static __device__ __forceinline__ void lds(const uint32_t j, const ResPos& ps, const double* up, const double* vp,
                                               d2 (&uu)[RES_UNROLL], d2 (&vv)[RES_UNROLL]) {
#pragma unroll
        for (int u = 0; u < RES_UNROLL; ++u) {
            const uint32_t p = ps.p_first + (ER + j * RES_UNROLL + u) * ps.p_stride;
            const uint32_t pc = p < ps.n2 ? p : 0u;  // (a valid address; the value is masked out where it is used)
            uu[u] = ld16<NT>(up, pc);
            if constexpr (NEEDV) vv[u] = ld16<NT>(vp, pc);
            else vv[u] = uu[u];
        }
    }
    // group GI of the step, whichever kind it is.  (Past the step's last group the LDS form loads pair 0 -- in cache,
    // unused -- rather than nothing: an unconditional fetch keeps the window free of merges of loaded and unloaded values.)
    template <int GI>
    static __device__ __forceinline__ void any(const ResPos& ps, const double* up, const double* vp, d2 (&uu)[RES_UNROLL],
                                               d2 (&vv)[RES_UNROLL]) {
        if constexpr (GI < NG) reg<GI>(ps, up, vp, uu, vv);
        else lds((uint32_t)(GI - NG), ps, up, vp, uu, vv);
    }
    template <int K = 0>
    static __device__ __forceinline__ void window(ResWin& w, const ResPos& ps, const double* up, const double* vp) {
        if constexpr (K < RES_AHEAD) {
            any<K>(ps, up, vp, w.u[K], w.v[K]);
            window<K + 1>(w, ps, up, vp);
        }
    }
};
__device__ __forceinline__ void res_shift(ResWin& w, const d2 (&fu)[RES_UNROLL], const d2 (&fv)[RES_UNROLL]) {
#pragma unroll
    for (int k = 0; k + 1 < RES_AHEAD; ++k)
#pragma unroll
        for (int u = 0; u < RES_UNROLL; ++u) { w.u[k][u] = w.u[k + 1][u]; w.v[k][u] = w.v[k + 1][u]; }
#pragma unroll
    for (int u = 0; u < RES_UNROLL; ++u) { w.u[RES_AHEAD - 1][u] = fu[u]; w.v[RES_AHEAD - 1][u] = fv[u]; }
}
// WO (the LAST step only, MODE 2): where the step leaves q.  0: on the chip, as every other step does (the plain case: d is
// written by a pass of stores after the last hand-off).  2: in d, under OWL-QN, with the orthant projection
// (orthantwise.rs:140-161) applied on the way out: the step streams v = pg anyway, so pg is not read a second time (25
// instead of 26 passes at m = 6); acc0 stays ||d||^2 BEFORE the projection (lbfgs.rs:543 precedes :554), acc2 / acc3 sum
// ||d||^2 and pg.d of the projected direction.  (The plain case stored the same way measured 1.5-3.6 % SLOWER than the
// separate pass at 3e6 ... 1.25e7 elements, profiles/r03_resident_fused_writeout_ab.log.)
struct ResOut {
    double* d;
    ResOwl ow;
};
template <int WO>
__device__ __forceinline__ void res_project_pair(d2& q, const d2 pg, const uint64_t gi, const ResOwl& ow, double* acc) {
    if constexpr (WO == 2) {
        q.x = res_project(q.x, pg.x, gi, ow);
        acc[2] += q.x * q.x; acc[3] += pg.x * q.x;
        q.y = res_project(q.y, pg.y, gi + 1, ow);
        acc[2] += q.y * q.y; acc[3] += pg.y * q.y;
    }
}
template <int MODE, int WO>
__device__ __forceinline__ void res_pin_step(double* acc) {
    res_pin_sums<(MODE == 2 ? 2 : 1)>(acc);
    if constexpr (WO == 2) res_pin_sums<2>(acc + 2);
}
template <int G, int ER, bool NT, int MODE, int WO>
struct ResRegStep {
    static constexpr int NG = ER / RES_UNROLL;
    static __device__ __forceinline__ void run(ResWin& w, const ResPos& ps, const double* up, const double* vp, const double c,
                                               const double gamma, double* acc, const ResOut& out) {
        if constexpr (G < NG) {
            d2 fu[RES_UNROLL], fv[RES_UNROLL];
            ResFetch<ER, NT, MODE != 1>::template any<G + RES_AHEAD>(ps, up, vp, fu, fv);
            // a compiler barrier: without it every group's loads are hoisted to the top of the step and the kernel spills
            asm volatile("" ::: "memory");
            d2 q0 = acc_get<G * RES_UNROLL + 0>(), q1 = acc_get<G * RES_UNROLL + 1>(), q2 = acc_get<G * RES_UNROLL + 2>(),
               q3 = acc_get<G * RES_UNROLL + 3>();
            res_one<MODE>(q0, w.u[0][0], w.v[0][0], c, gamma, acc);
            res_one<MODE>(q1, w.u[0][1], w.v[0][1], c, gamma, acc);
            res_one<MODE>(q2, w.u[0][2], w.v[0][2], c, gamma, acc);
            res_one<MODE>(q3, w.u[0][3], w.v[0][3], c, gamma, acc);
            if constexpr (WO == 0) {
                acc_put<G * RES_UNROLL + 0>(q0);
                acc_put<G * RES_UNROLL + 1>(q1);
                acc_put<G * RES_UNROLL + 2>(q2);
                acc_put<G * RES_UNROLL + 3>(q3);
            } else {  // the last step: q leaves the chip here
                const uint32_t o0 = res_opaque(ps.p_first * 16u), os = ps.p_stride * 16u;
                const uint32_t oa = o0 + (uint32_t)(G * RES_UNROLL + 0) * os, ob = o0 + (uint32_t)(G * RES_UNROLL + 1) * os,
                               oc = o0 + (uint32_t)(G * RES_UNROLL + 2) * os, od = o0 + (uint32_t)(G * RES_UNROLL + 3) * os;
                res_project_pair<WO>(q0, w.v[0][0], out.ow.gofs + 2ull * (oa >> 4), out.ow, acc);
                res_project_pair<WO>(q1, w.v[0][1], out.ow.gofs + 2ull * (ob >> 4), out.ow, acc);
                res_project_pair<WO>(q2, w.v[0][2], out.ow.gofs + 2ull * (oc >> 4), out.ow, acc);
                res_project_pair<WO>(q3, w.v[0][3], out.ow.gofs + 2ull * (od >> 4), out.ow, acc);
                st16_at<NT>(out.d, oa, q0);
                st16_at<NT>(out.d, ob, q1);
                st16_at<NT>(out.d, oc, q2);
                st16_at<NT>(out.d, od, q3);
            }
            res_pin_step<MODE, WO>(acc);
            res_shift(w, fu, fv);
            ResRegStep<G + 1, ER, NT, MODE, WO>::run(w, ps, up, vp, c, gamma, acc, out);
        }
    }
};
// one whole step; on entry the window holds the operands of groups 0 .. RES_AHEAD-1, on exit nothing.
template <int ER, bool NT, int MODE, int WO = 0>
__device__ __forceinline__ void res_step(ResWin& w, d2* q_lds, const ResPos& ps, const double* up, const double* vp, const double c,
                                         const double gamma, double* acc, const ResOut& out = ResOut{}) {
    static_assert(WO == 0 || MODE == 2, "only the last step writes q out");
    ResRegStep<0, ER, NT, MODE, WO>::run(w, ps, up, vp, c, gamma, acc, out);
    const uint32_t tid = threadIdx.x, NL = ps.EL / RES_UNROLL;
    for (uint32_t j = 0; j < NL; ++j) {
        d2 fu[RES_UNROLL], fv[RES_UNROLL];
        ResFetch<ER, NT, MODE != 1>::lds(j + RES_AHEAD, ps, up, vp, fu, fv);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < RES_UNROLL; ++u) {
            const uint32_t p = ps.p_first + (ER + j * RES_UNROLL + u) * ps.p_stride;
            d2 uu = w.u[0][u], vv = w.v[0][u];
            if (!(p < ps.n2)) { uu = d2{0.0, 0.0}; vv = d2{0.0, 0.0}; }  // padding pairs stay 0 and add 0
            d2 q = q_lds[(size_t)(j * RES_UNROLL + u) * BLOCK + tid];
            res_one<MODE>(q, uu, vv, c, gamma, acc);
            if constexpr (WO == 0) {
                q_lds[(size_t)(j * RES_UNROLL + u) * BLOCK + tid] = q;
            } else {  // the last step: q leaves the chip here (padding pairs: q = 0, nothing to store)
                res_project_pair<WO>(q, vv, out.ow.gofs + 2ull * p, out.ow, acc);
                if (p < ps.n2) st16<NT>(out.d, p, q);
            }
        }
        res_pin_step<MODE, WO>(acc);
        res_shift(w, fu, fv);
    }
}

// ---- hybrid: shards too large for the chip.  Rounds e >= ER + EL of a thread keep their part of q in HBM -- in `d`,
// as the kernel-per-step path does -- and every step streams them 3r + 1w; the first ER + EL rounds stay on the chip
// (2r).  The same thread reads and rewrites the same pairs of `d` in every step, so nothing new crosses workgroups.
// In the FIRST step the source of q is g (negated on the fly: the kernel-per-step path's fused first step), so `d` is
// never initialised by a pass of its own.  Loads of the next group of RES_UNROLL pairs are in flight while one is
// worked on (2 x 12 loads of 16 bytes per thread).
// one pair of a hybrid round.  MODE 3: the last step under OWL-QN, projection included (ops.h VMODE 3, element by element:
// acc0 += q.q before the projection -- lbfgs.rs:543 precedes :554 --, then acc2 += q.q and acc3 += pg.q after it)
template <int MODE>
__device__ __forceinline__ void res_hbm_one(d2& q, const d2 uu, const d2 vv, const double c, const double gamma, double* acc,
                                            const ResOwl& ow, const uint32_t pair) {
    if constexpr (MODE == 3) {
        const uint64_t gi = ow.gofs + 2ull * pair;
        q.x = q.x + c * uu.x;
        q.y = q.y + c * uu.y;
        acc[0] += q.x * q.x;
        q.x = res_project(q.x, vv.x, gi, ow);
        acc[2] += q.x * q.x; acc[3] += vv.x * q.x;
        acc[0] += q.y * q.y;
        q.y = res_project(q.y, vv.y, gi + 1, ow);
        acc[2] += q.y * q.y; acc[3] += vv.y * q.y;
    } else {
        res_one<MODE>(q, uu, vv, c, gamma, acc);
    }
}
template <int MODE>
__device__ __forceinline__ void res_hbm_pin(double* acc) {
    if constexpr (MODE == 3) { res_pin(acc[0]); res_pin(acc[2]); res_pin(acc[3]); }
    else res_pin_sums<(MODE == 2 ? 2 : 1)>(acc);
}
struct ResHbmGroup {
    d2 q[RES_HBM_UNROLL], u[RES_HBM_UNROLL], v[RES_HBM_UNROLL];
};
// (full rounds only: no bounds checks, no branches)
template <bool NT, bool QNT, bool NEEDV>
__device__ __forceinline__ void res_hbm_fetch(ResHbmGroup& g, const uint32_t e, const ResPos& ps, const double* qsrc,
                                              const double* up, const double* vp) {
    const uint32_t p0 = res_opaque(ps.p_first + e * ps.p_stride);
#pragma unroll
    for (int k = 0; k < RES_HBM_UNROLL; ++k) {
        const uint32_t p = p0 + (uint32_t)k * ps.p_stride;
        g.q[k] = ld16<QNT>(qsrc, p);
        g.u[k] = ld16<NT>(up, p);
        if constexpr (NEEDV) g.v[k] = ld16<NT>(vp, p);
        else g.v[k] = g.u[k];
    }
}
template <bool NT, bool QNT, int MODE>
__device__ __forceinline__ void res_hbm_work(ResHbmGroup& g, const uint32_t e, const ResPos& ps, const double qsign, double* d,
                                             const double c, const double gamma, double* acc, const ResOwl& ow) {
    const uint32_t p0 = res_opaque(ps.p_first + e * ps.p_stride);
#pragma unroll
    for (int k = 0; k < RES_HBM_UNROLL; ++k) {
        d2 q = g.q[k];
        q.x = q.x * qsign;  // +1.0, or -1.0 in the first step (q = -g: exact)
        q.y = q.y * qsign;
        const uint32_t p = p0 + (uint32_t)k * ps.p_stride;
        res_hbm_one<MODE>(q, g.u[k], g.v[k], c, gamma, acc, ow, p);
        st16<QNT>(d, p, q);
    }
    res_hbm_pin<MODE>(acc);
}
// rounds [e_begin, e_end) of this thread; every round before the shard's last one is full for every thread, so groups of
// RES_HBM_UNROLL rounds below `e_full` run without checks and the (at most RES_HBM_UNROLL) rounds after them one at a
// time.  `rev`: sweep from the high addresses down (the groups in descending order, the last rounds first).
template <bool NT, bool QNT, int MODE>
__device__ __forceinline__ void res_hbm_tail(const uint32_t e_full, const uint32_t e_end, const ResPos& ps, const double* qsrc,
                                             const double qsign, double* d, const double* up, const double* vp, const double c,
                                             const double gamma, double* acc, const ResOwl& ow) {
    for (uint32_t e = e_full; e < e_end; ++e) {  // the last few rounds, the ragged one among them
        const uint32_t p = ps.p_first + e * ps.p_stride;
        if (p < ps.n2) {
            d2 q = ld16<QNT>(qsrc, p);
            const d2 uu = ld16<NT>(up, p);
            d2 vv = uu;
            if constexpr (MODE != 1) vv = ld16<NT>(vp, p);
            q.x = q.x * qsign;
            q.y = q.y * qsign;
            res_hbm_one<MODE>(q, uu, vv, c, gamma, acc, ow, p);
            st16<QNT>(d, p, q);
        }
    }
}
template <bool NT, bool QNT, int MODE>
__device__ __forceinline__ void res_step_hbm(const uint32_t e_begin, const uint32_t e_end, const bool rev, const ResPos& ps,
                                             const double* qsrc, const double qsign, double* d, const double* up, const double* vp,
                                             const double c, const double gamma, double* acc, const ResOwl& ow) {
    if (e_begin >= e_end) return;
    asm volatile("" ::: "memory");
    const uint32_t ngroups = (e_end - 1 - e_begin) / RES_HBM_UNROLL;  // groups made of full rounds only
    const uint32_t e_full = e_begin + ngroups * RES_HBM_UNROLL;
    if (rev) res_hbm_tail<NT, QNT, MODE>(e_full, e_end, ps, qsrc, qsign, d, up, vp, c, gamma, acc, ow);
    if (ngroups > 0) {
        ResHbmGroup ga, gb;  // two buffers, used alternately: the next group's 12 loads fly while one is worked on
        const uint32_t last = ngroups - 1;
        auto first_round = [&](const uint32_t i) {  // of the i-th group in sweep order (past the end: the last one again, unused)
            const uint32_t k = min(i, last);
            return e_begin + (rev ? last - k : k) * RES_HBM_UNROLL;
        };
        res_hbm_fetch<NT, QNT, MODE != 1>(ga, first_round(0), ps, qsrc, up, vp);
        uint32_t i = 0;
        for (;;) {
            res_hbm_fetch<NT, QNT, MODE != 1>(gb, first_round(i + 1), ps, qsrc, up, vp);
            asm volatile("" ::: "memory");
            res_hbm_work<NT, QNT, MODE>(ga, first_round(i), ps, qsign, d, c, gamma, acc, ow);
            if (++i >= ngroups) break;
            res_hbm_fetch<NT, QNT, MODE != 1>(ga, first_round(i + 1), ps, qsrc, up, vp);
            asm volatile("" ::: "memory");
            res_hbm_work<NT, QNT, MODE>(gb, first_round(i), ps, qsign, d, c, gamma, acc, ow);
            if (++i >= ngroups) break;
        }
    }
    if (!rev) res_hbm_tail<NT, QNT, MODE>(e_full, e_end, ps, qsrc, qsign, d, up, vp, c, gamma, acc, ow);
    asm volatile("" ::: "memory");
}

// The HBM rounds of one step.  Rounds [e0, ep) -- the "cache slice" of q -- are read and written with the DEFAULT cache
// policy, rounds [ep, e1) with the `nt` hint like the history vectors.  MI355X's 256 MiB Infinity Cache sits between the
// L2s and HBM: what carries `nt` passes through it without staying, so a slice of q of about its size that is NOT marked
// `nt` is still there when the next step reads it again -- and is rewritten there -- instead of making the round trip to
// HBM (ResArgs::hbm_plain_pairs; 0 = everything streamed).  Measured (profiles/r03_hybrid_cache_slice_ab.log): the
// two-loop kernel at n = 1e8, m = 10 9.71 -> 9.26 ms with a 256 MiB slice and the alternating sweep; 5e7: 4.58 -> 4.07;
// 2.5e7: 1.98 -> 1.80; a slice beyond 256 MiB without the alternating sweep thrashes (10.0 ms).
template <bool NT, int MODE>
__device__ __forceinline__ void res_hbm_rounds(const uint32_t e0, const uint32_t ep, const uint32_t e1, const bool rev, const ResPos& ps,
                                               const double* qsrc, const double qsign, double* d, const double* up, const double* vp,
                                               const double c, const double gamma, double* acc, const ResOwl& ow) {
    if (rev) {
        res_step_hbm<NT, NT, MODE>(ep, e1, true, ps, qsrc, qsign, d, up, vp, c, gamma, acc, ow);
        if constexpr (NT) res_step_hbm<NT, false, MODE>(e0, ep, true, ps, qsrc, qsign, d, up, vp, c, gamma, acc, ow);
    } else {
        if constexpr (NT) res_step_hbm<NT, false, MODE>(e0, ep, false, ps, qsrc, qsign, d, up, vp, c, gamma, acc, ow);
        res_step_hbm<NT, NT, MODE>(ep, e1, false, ps, qsrc, qsign, d, up, vp, c, gamma, acc, ow);
    }
}

// ER = pairs per thread held in (accumulation) registers: a multiple of RES_UNROLL, at most 60.
// HYB: the shard is larger than the chip (a.hbm_pairs > 0; ER = 60 only): the hybrid rounds are compiled in.
template <int ER, bool NT, bool HYB>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(1, 1)))
void two_loop_resident_kernel(const ResArgs a, const RedCtl red) {
    static_assert(ER % RES_UNROLL == 0 && ER >= 0 && 4 * ER <= 256, "pairs in AGPRs (ER = 0: everything in LDS)");
    asm volatile("" ::: "a255");  // this kernel owns the whole accumulation register file (the wave gets 512 registers)
    extern __shared__ d2 q_lds[];  // [lds_pairs][BLOCK]
    __shared__ double lds[8][WAVES];  // (res_exchange: rows 0-3 and 4-7)
    __shared__ double s_tot[4];
    __shared__ double s_alpha[RES_MAX_STEPS / 2];
    __shared__ unsigned int s_bits[P2P_MAX_WORLD][MAX_RED][2];  // (p2p_exchange's staging: 4 KiB; unused with one rank)
    constexpr int NG = ER / RES_UNROLL;
    if (a.fault_drop_last && blockIdx.x == gridDim.x - 1 && gridDim.x > 1) return;  // (uniform per workgroup; see ResArgs)
    const DevCounters c0 = load_counters(red);
    // 32-bit pair indices (the host admits only shards of < 2^28 pairs): one uniform stride, one add per element
    const uint32_t n2 = (uint32_t)(a.n >> 1);
    const uint32_t G = gridDim.x, B = blockIdx.x, tid = threadIdx.x;
    const uint32_t EL = a.lds_pairs;
    const uint32_t p_first = B * BLOCK + tid, p_stride = G * BLOCK;
    // pair e of this thread: e < ER in registers (all in range), ER <= e < ER + EL in LDS; global pair (e*G + B)*BLOCK + tid
    const bool tail_owner = (a.n & 1) && B == G - 1 && tid == 0;  // odd n: the last element, kept by one thread
    double q_tail = 0.0;

    double acc[4] = {0.0, 0.0, 0.0, 0.0};  // (acc2, acc3: the projected direction's sums, OWL-QN's last step and write-out only)
    const ResOwl ow{a.gofs, a.owl_start, a.owl_end};
    // ---- q = -g ; optionally the first numerator s.(-g)
    const bool need_first = a.first_dot == nullptr;
    const uint32_t b_first = p_first * 16u, b_stride = p_stride * 16u;  // byte offsets (shards of < 2^28 pairs: < 2^32 bytes)
    if (need_first) ResGroups<0, NG, NT, 0>::template init<true>(b_first, b_stride, a.g, a.first_s, acc);
    else ResGroups<0, NG, NT, 0>::template init<false>(b_first, b_stride, a.g, a.first_s, acc);
    for (uint32_t e0 = 0; e0 < EL; e0 += RES_UNROLL) {
#pragma unroll
        for (int u = 0; u < RES_UNROLL; ++u) {
            const uint32_t p = p_first + (ER + e0 + u) * p_stride;
            d2 q = d2{0.0, 0.0}, sv = d2{0.0, 0.0};
            if (p < n2) {
                const d2 gv = ld16<NT>(a.g, p);
                q.x = -gv.x;
                q.y = -gv.y;
                if (need_first) sv = ld16<NT>(a.first_s, p);
            }
            acc[0] += sv.x * q.x;
            acc[0] += sv.y * q.y;
            q_lds[(size_t)(e0 + u) * BLOCK + tid] = q;
        }
    }
    // hybrid rounds: q stays in HBM (the first step reads it from g); only the first numerator needs them here
    const uint32_t EH0 = ER + EL, EH1 = ER + EL + (HYB ? a.hbm_pairs : 0u);
    const uint32_t EHP = NT ? EH0 + (HYB ? min(a.hbm_plain_pairs, a.hbm_pairs) : 0u) : EH0;  // (see res_hbm_rounds)
    if (HYB && need_first) {
        for (uint32_t e = EH0; e < EH1; ++e) {
            const uint32_t p = p_first + e * p_stride;
            if (p < n2) {
                const d2 gv = ld16<NT>(a.g, p), sv = ld16<NT>(a.first_s, p);
                acc[0] += sv.x * (-gv.x);
                acc[0] += sv.y * (-gv.y);
            }
        }
    }
    if (tail_owner) {
        q_tail = -a.g[a.n - 1];
        if (need_first) acc[0] += a.first_s[a.n - 1] * q_tail;
    }
    unsigned int tag = c0.red_epoch, ptag = c0.p2p_epoch;
    int parity = 0;
    double dot;
    // (see ResWin: the next step's first groups are loaded BEFORE the hand-off that yields its coefficient, so HBM keeps
    // streaming while the workgroups wait -- a wave's loads return in order, the polls of the hand-off queue behind them)
    ResWin win;
    const ResPos ps{p_first, p_stride, n2, EL};
    auto prefetch = [&](const int si) {
        const double* up = a.step[si].u;
        const double* vp = a.step[si].v ? a.step[si].v : up;  // (the gamma transition has no v: those loads go unused)
        ResFetch<ER, NT, true>::window(win, ps, up, vp);
    };
    prefetch(0);
    // (see TOUCHING at the top) the depth is a property of the instantiation: 16 rounds where a thread owns >= 61 rounds (ER = 60),
    // 8 below -- the host asks for exactly that (a.touch_rounds), so no toucher issues a load for a line it has just requested
    // (round 4 instantiated 16 everywhere and clamped: half of a toucher's loads were repeats on the shorter shards)
    constexpr int TOUCH = (HYB || ER == 0) ? 0 : (ER == 60 ? LH_RES_TOUCH : LH_RES_TOUCH / 2);
    ResSink sink{};
    ResXchgAcc xacc;
    // (LH_RES_TOUCH) what a workgroup touches while it waits in the hand-off BEFORE step `si`: the rounds behind the window
    ResTouch tch;
    tch.first = B * (uint32_t)(BLOCK * 16);
    tch.round_stride = G * (uint32_t)(BLOCK * 16);
    tch.r0 = (uint32_t)(RES_AHEAD * RES_UNROLL);
    // (rounds beyond the depth asked for at run time, or beyond the thread's last round, fall back on the last one touched: lines
    // that are in the L2 already.  touch_rounds = 0 touches one round: next to nothing)
    tch.r_end = min(tch.r0 + max(min(a.touch_rounds, (uint32_t)(TOUCH > 0 ? TOUCH : 1)), 1u), max(a.pairs_per_thread, 1u));
    tch.limit = (uint32_t)(a.n * 8ull - 8ull);
    auto touch_for = [&](const int si) {
        ResTouch t = tch;
        const int sj = si < a.nsteps ? si : a.nsteps - 1;  // (past the last step: its operands again -- cached, harmless)
        t.u = reinterpret_cast<const char*>(a.step[sj].u);
        t.v = reinterpret_cast<const char*>(a.step[sj].v ? a.step[sj].v : a.step[sj].u);
        return t;
    };
    if (need_first) {
        double t1[1] = {acc[0]};
        res_exchange<1, TOUCH>(t1, red, tag, ptag, parity, lds, s_tot, s_bits, sink, xacc, touch_for(0));
        dot = t1[0];
        tag = next_epoch(tag);
        ptag = next_epoch(ptag);
        parity ^= 1;
    } else {
        dot = *a.first_dot;
    }
    const double gamma = *a.gnum / *a.gden;  // lbfgs.rs:691 (the quotient every path forms)

    // ---- the 2*bound steps
    for (int si = 0; si < a.nsteps; ++si) {
        const ResStep st = a.step[si];
        const double r = dot / *st.ys;
        double c;
        if (st.mode_b) {
            c = s_alpha[st.alpha_idx] - r;
        } else {
            c = -r;
            if (tid == 0) s_alpha[st.alpha_idx] = r;  // read again (by every thread) only after later barriers
            if (B == 0 && tid == 0) *st.alpha = r;
        }
        acc[0] = 0.0;
        acc[1] = 0.0;
        acc[2] = 0.0;
        acc[3] = 0.0;
        const int mode = st.last ? 2 : (st.v == nullptr ? 1 : 0);
        const double* qsrc = si == 0 ? a.g : a.d;  // (hybrid rounds only)
        const double qsign = si == 0 ? -1.0 : 1.0;
        // hybrid: odd steps sweep the HBM rounds from the top down (what a step wrote last, the next reads first).  Pays
        // together with the cache slice of res_hbm_rounds: a slice larger than the Infinity Cache then degrades gracefully
        // instead of thrashing (profiles/r03_hybrid_cache_slice_ab.log)
        const bool rev = (si & 1) != 0;
        if (mode == 0) {
            res_step<ER, NT, 0>(win, q_lds, ps, st.u, st.v, c, gamma, acc);
            if constexpr (HYB) res_hbm_rounds<NT, 0>(EH0, EHP, EH1, rev, ps, qsrc, qsign, a.d, st.u, st.v, c, gamma, acc, ow);
        } else if (mode == 1) {
            res_step<ER, NT, 1>(win, q_lds, ps, st.u, st.u, c, gamma, acc);
            if constexpr (HYB) res_hbm_rounds<NT, 1>(EH0, EHP, EH1, rev, ps, qsrc, qsign, a.d, st.u, st.u, c, gamma, acc, ow);
        } else {  // the last step: under OWL-QN q goes out to d as it is completed, projected on the way
            const ResOut out{a.d, ow};
            if (a.owl) res_step<ER, NT, 2, 2>(win, q_lds, ps, st.u, st.v, c, gamma, acc, out);
            else res_step<ER, NT, 2, 0>(win, q_lds, ps, st.u, st.v, c, gamma, acc, out);
            if constexpr (HYB) {
                if (a.owl) res_hbm_rounds<NT, 3>(EH0, EHP, EH1, rev, ps, qsrc, qsign, a.d, st.u, st.v, c, gamma, acc, ow);  // (projected here)
                else res_hbm_rounds<NT, 2>(EH0, EHP, EH1, rev, ps, qsrc, qsign, a.d, st.u, st.v, c, gamma, acc, ow);
            }
        }
        if constexpr (TOUCH > 0) res_touch_retire(sink);  // (what the hand-off before this step touched has long returned)
        if (si + 1 < a.nsteps) prefetch(si + 1);
        if (tail_owner) {
            const double ut = st.u[a.n - 1], vt = (mode == 1) ? ut : st.v[a.n - 1];
            q_tail = q_tail + c * ut;
            if (mode == 1) q_tail = q_tail * gamma;
            if (mode == 2) { acc[0] += q_tail * q_tail; acc[1] += vt * q_tail; }
            else acc[0] += vt * q_tail;
        }
        if (mode == 2) {
            // OWL-QN: the last step's sums feed no coefficient -- they travel with the projected direction's sums in ONE
            // hand-off of four values after the write-out below (the launch-per-step path's last kernel also closes
            // its four sums in one reduction: ranks that take different paths still exchange the same sequence)
            if (a.owl) break;
            double t2[2] = {acc[0], acc[1]};
            res_exchange<2, TOUCH>(t2, red, tag, ptag, parity, lds, s_tot, s_bits, sink, xacc, touch_for(si));
            acc[0] = t2[0];
            acc[1] = t2[1];
        } else {
            double t1[1] = {acc[0]};
            res_exchange<1, TOUCH>(t1, red, tag, ptag, parity, lds, s_tot, s_bits, sink, xacc, touch_for(si + 1));
            dot = t1[0];
        }
        tag = next_epoch(tag);
        ptag = next_epoch(ptag);
        parity ^= 1;
    }

    // ---- OWL-QN: d has been written by the last step (res_step WO = 2); what is left: the odd element and the hand-off of
    // the four sums (||d||^2 before the projection, --, ||d||^2 and pg.d after it).  Plain: d = q, a pass of stores.
    if (a.owl) {
        const double* pgp = a.step[a.nsteps - 1].v;  // the last step's v is pg
        if (tail_owner) {
            const double pgt = pgp[a.n - 1];
            q_tail = res_project(q_tail, pgt, a.gofs + a.n - 1, ow);
            acc[2] += q_tail * q_tail; acc[3] += pgt * q_tail;
            a.d[a.n - 1] = q_tail;
        }
        double t4[4] = {acc[0], 0.0, acc[2], acc[3]};  // (the layout of ops.h VMODE 3: slot 1 is unused there too)
        res_exchange<4, TOUCH>(t4, red, tag, ptag, parity, lds, s_tot, s_bits, sink, xacc, touch_for(a.nsteps));
        tag = next_epoch(tag);
        ptag = next_epoch(ptag);
        acc[0] = t4[0];
        acc[1] = t4[2];
        acc[2] = t4[3];
    } else {
        // (LH_RES_D_NT=0, A/B builds: d written with the default cache policy -- the line search reads it next and a shard's d fits
        // the Infinity Cache; measured twice, rounds 4 and 6: what the trials gain the two-loop kernel loses, EXPERIMENTS.md)
        constexpr bool DNT = NT && (LH_RES_D_NT != 0);
        ResGroups<0, NG, DNT, 0>::store(b_first, b_stride, a.d);
        for (uint32_t e = 0; e < EL; ++e) {
            const uint32_t p = p_first + (ER + e) * p_stride;
            if (p < n2) st16<DNT>(a.d, p, q_lds[(size_t)e * BLOCK + tid]);
        }
        if (tail_owner) a.d[a.n - 1] = q_tail;
    }
    if (B == 0 && tid == 0) {
        // plain: ||d||^2, g.d.   OWL-QN: ||d||^2 before the projection, (unused), ||d||^2 and pg.d after it -- the layout of
        // the launch-per-step last step (ops.h VMODE 3)
        const int nout = a.owl ? 4 : 2;
        double outv[4] = {acc[0], acc[1], 0.0, 0.0};
        if (a.owl) { outv[1] = 0.0; outv[2] = acc[1]; outv[3] = acc[2]; }
        for (int k = 0; k < nout; ++k) a.out_dn[k] = outv[k];
        red.ctr->red_epoch = tag;  // every workgroup has left the last hand-off with `tag - 1`: nobody reads the counter again
        if (p2p_active(red.p2p)) {
            red.ctr->p2p_epoch = ptag;
            DevXchg* const xs = dev_xchg(red.ctr, 1u);  // (this kernel's exchanges all belong to a two-loop)
            DevXchg x = *xs;
            x.p2p_ticks += xacc.p2p_ticks;
            x.local_ticks += xacc.local_ticks;
            x.count += xacc.count;
            *xs = x;
        }
        if (red.mirror.host_board) {
            const long i0 = a.out_dn - red.mirror.board;
            if (i0 >= 0 && i0 + nout - 1 < red.mirror.slots) {
                for (int k = 0; k < nout; ++k)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(red.mirror.host_board + i0 + k),
                                       (unsigned long long)__double_as_longlong(outv[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (red.mirror.host_err) {
                const unsigned int e = __hip_atomic_load(red.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (e) __hip_atomic_store(red.mirror.host_err, (unsigned long long)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const unsigned long long seq = c0.mirror_seq + 1ull;
            red.ctr->mirror_seq = seq;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(red.mirror.host_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace lh
