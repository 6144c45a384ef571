// rust-lbfgs_amd/csrc/context.hip -- the context of include/lbfgs_hip.h: device, stream, scalar board, reduction scratch,
// communicators (lazy RCCL, P2P mailboxes in device or host memory, host callback), pooled uncached memory, device vectors,
// scalar reads with the recovery from a timed-out resident two-loop, buffers, event timing.  No vector kernels here (they
// are launched from lbfgs_hip.hip); the only kernel is the stand-alone P2P exchange of host-produced sums.
//
// There is deliberately NO CPU fallback: without a GPU lbfgs_hip_ctx_create() fails with LBFGS_HIP_ERR_NO_DEVICE.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <thread>

#include "ctx.h"
#include "ext_protocol.h"

// ------------------------------------------------------------------------------------ RCCL (lazy)
// RCCL is only needed when world > 1, so it is dlopen'ed on first use; a single-GPU process
// never loads it.
namespace {
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id_t*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id_t, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*CommCount)(const nccl_comm_t, int*) = nullptr;     // what RCCL itself says the communicator spans ...
    int (*CommUserRank)(const nccl_comm_t, int*) = nullptr;  // ... and who this process is in it
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*CommAbort)(void*) = nullptr;  // optional
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
const int kNcclDouble = 8, kNcclSum = 0;

bool rccl_load(std::string* err) {
    if (g_rccl.ok) return true;
    // LBFGS_HIP_RCCL: the RCCL build that belongs to the HIP runtime in use (set by the Python loader when it
    // pre-loaded PyTorch's bundled runtime); otherwise whatever librccl.so.1 the process already has / finds
    const char* names[] = {getenv("LBFGS_HIP_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        *err = std::string("cannot dlopen librccl: ") + dlerror();
        return false;
    }
#define LH_SYM(field, name)                                                        \
    *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, name);                         \
    if (!g_rccl.field) { *err = std::string("librccl lacks ") + name; return false; }
    LH_SYM(GetUniqueId, "ncclGetUniqueId")
    LH_SYM(CommInitRank, "ncclCommInitRank")
    LH_SYM(CommDestroy, "ncclCommDestroy")
    LH_SYM(CommCount, "ncclCommCount")
    LH_SYM(CommUserRank, "ncclCommUserRank")
    LH_SYM(AllReduce, "ncclAllReduce")
    LH_SYM(GroupStart, "ncclGroupStart")
    LH_SYM(GroupEnd, "ncclGroupEnd")
    LH_SYM(GetErrorString, "ncclGetErrorString")
#undef LH_SYM
    *(void**)(&g_rccl.CommAbort) = dlsym(g_rccl.handle, "ncclCommAbort");  // (absent: a communicator that cannot drain is left alone)
    g_rccl.ok = true;
    return true;
}
std::string g_create_error;  // last error of a failed ctx_create (no ctx to hold it)

#ifndef LBFGS_HIP_BUILD_ID
#define LBFGS_HIP_BUILD_ID "unstamped"
#endif
// (the marker makes the id readable from the FILE, without loading it: rust-lbfgs_amd/_build.py embedded_id)
const char lbfgs_hip_build_id_marker[] = "LBFGS_HIP_BUILD_ID=" LBFGS_HIP_BUILD_ID;

// ---- HOST-placed P2P mailboxes (lbfgs_hip.h): a POSIX shared-memory segment, registered with HIP so that kernels reach
// it through a device address (fine-grained host memory: system-scope atomics go straight to host DRAM over PCIe).
// The 64-byte handle carries a magic word and the segment's name instead of a hipIpcMemHandle_t.
constexpr char HOST_MBOX_MAGIC[8] = {'L', 'H', 'M', 'B', 'O', 'X', 'H', '1'};
struct HostMbox {
    void* host = nullptr;
    size_t bytes = 0;
    std::string name;
    bool owner = false;   // this process created the segment: it unlinks the name
    bool linked = false;  // the name still exists
};
std::map<void*, HostMbox> g_host_mbox;  // device address -> mapping
std::mutex g_host_mbox_mu;

// uncached blocks that no context is using, per device: granule buffers (lbfgs_hip_ctx_create) and device-placed P2P
// mailboxes.  Uncached memory is never handed back to the allocator (see lbfgs_hip_ctx_create).
std::map<int, std::vector<void*>> g_uc_pool, g_uc_mbox_pool;
std::mutex g_uc_pool_mu;
void uc_mbox_retire(int device, void* p) {
    std::lock_guard<std::mutex> lk(g_uc_pool_mu);
    g_uc_mbox_pool[device].push_back(p);
}
// Peers' device-placed mailboxes stay mapped for the life of the process, for the same reason: a peer that pools its
// mailbox exports the same IPC handle again, and an address range that was mapped uncached is not unmapped and handed
// to the allocator for something else.  Keyed by the 64 handle bytes.
std::map<std::string, void*> g_ipc_open;
hipError_t ipc_open_cached(const hipIpcMemHandle_t& hdl, void** out) {
    const std::string key(reinterpret_cast<const char*>(&hdl), sizeof(hdl));
    std::lock_guard<std::mutex> lk(g_uc_pool_mu);
    auto it = g_ipc_open.find(key);
    if (it != g_ipc_open.end()) {
        *out = it->second;
        return hipSuccess;
    }
    const hipError_t e = hipIpcOpenMemHandle(out, hdl, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) g_ipc_open[key] = *out;
    return e;
}

// Device-placed mailboxes created by THIS process, keyed by the 64 bytes of the IPC handle they export.  A process that hosts
// several ranks (several contexts, one host thread each) reaches a same-process peer's mailbox through the pointer itself:
// hipIpcOpenMemHandle refuses handles of the opening process's own allocations.  Entries stay for the life of the process (a
// retired mailbox is pooled, never freed, and exports the same handle when it is used again).
struct LocalMbox {
    void* p;
    int device;
};
std::map<std::string, LocalMbox> g_local_mbox;
void local_mbox_register(const hipIpcMemHandle_t& hdl, void* p, int device) {
    std::lock_guard<std::mutex> lk(g_uc_pool_mu);
    g_local_mbox[std::string(reinterpret_cast<const char*>(&hdl), sizeof(hdl))] = LocalMbox{p, device};
}
bool local_mbox_lookup(const hipIpcMemHandle_t& hdl, LocalMbox* out) {
    std::lock_guard<std::mutex> lk(g_uc_pool_mu);
    auto it = g_local_mbox.find(std::string(reinterpret_cast<const char*>(&hdl), sizeof(hdl)));
    if (it == g_local_mbox.end()) return false;
    *out = it->second;
    return true;
}

void host_mbox_unlink_all() {  // atexit: names of segments this process still owns
    std::lock_guard<std::mutex> lk(g_host_mbox_mu);
    for (auto& kv : g_host_mbox)
        if (kv.second.owner && kv.second.linked) {
            (void)shm_unlink(kv.second.name.c_str());
            kv.second.linked = false;
        }
}

// map + register the segment `name` (create = this process makes it); -> device address or nullptr (*err set)
void* host_mbox_map(const std::string& name, size_t bytes, bool create, std::string* err) {
    const int fd = shm_open(name.c_str(), create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) {
        *err = "shm_open(" + name + "): " + strerror(errno);
        return nullptr;
    }
    if (create && ftruncate(fd, (off_t)bytes) != 0) {
        *err = std::string("ftruncate: ") + strerror(errno);
        (void)close(fd);
        (void)shm_unlink(name.c_str());
        return nullptr;
    }
    void* host = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    (void)close(fd);
    if (host == MAP_FAILED) {
        *err = std::string("mmap: ") + strerror(errno);
        if (create) (void)shm_unlink(name.c_str());
        return nullptr;
    }
    if (create) memset(host, 0, bytes);  // tag 0 is never a valid epoch
    void* dev = nullptr;
    hipError_t e = hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e == hipSuccess) {
        e = hipHostGetDevicePointer(&dev, host, 0);
        if (e != hipSuccess) (void)hipHostUnregister(host);
    }
    if (e != hipSuccess) {
        *err = std::string("hipHostRegister of the shared mailbox: ") + hipGetErrorString(e);
        (void)munmap(host, bytes);
        if (create) (void)shm_unlink(name.c_str());
        return nullptr;
    }
    static bool hooked = false;
    std::lock_guard<std::mutex> lk(g_host_mbox_mu);
    if (!hooked) {
        atexit(host_mbox_unlink_all);
        hooked = true;
    }
    HostMbox hm;
    hm.host = host;
    hm.bytes = bytes;
    hm.name = name;
    hm.owner = create;
    hm.linked = create;
    g_host_mbox[dev] = hm;
    return dev;
}
// -> true if `dev` was a host-placed mailbox (now released)
bool host_mbox_release(void* dev) {
    HostMbox hm;
    {
        std::lock_guard<std::mutex> lk(g_host_mbox_mu);
        auto it = g_host_mbox.find(dev);
        if (it == g_host_mbox.end()) return false;
        hm = it->second;
        g_host_mbox.erase(it);
    }
    (void)hipHostUnregister(hm.host);
    (void)munmap(hm.host, hm.bytes);
    if (hm.owner && hm.linked) (void)shm_unlink(hm.name.c_str());
    return true;
}
}  // namespace

// ------------------------------------------------------------------------------------ P2P all-reduce
// Stand-alone form of lh::p2p_exchange (stream.h) for sums the HOST produced (lbfgs_hip_scalars_allreduce);
// sums produced by a kernel are exchanged by that kernel's last workgroup and never come here.
struct P2PArgs {
    P2PCtl ctl;
    DevCounters* ctr;
    double* val[MAX_RED];
    int count;
    unsigned int xchg_class;
};

__global__ __launch_bounds__(64) void p2p_allreduce_kernel(const P2PArgs a) {
    __shared__ double vals[MAX_RED];
    __shared__ unsigned int bits[P2P_MAX_WORLD][MAX_RED][2];
    if ((int)threadIdx.x < a.count) vals[threadIdx.x] = *a.val[threadIdx.x];
    __syncthreads();
    const unsigned int epoch = a.ctr->p2p_epoch;
    DevXchg* const xs = dev_xchg(a.ctr, a.xchg_class);
    DevXchg x{};
    long long t1 = 0;
    if (threadIdx.x == 0) {
        x = *xs;
        t1 = wall_clock64();
    }
    p2p_exchange(a.ctl, epoch, vals, a.count, bits);
    if ((int)threadIdx.x < a.count) *a.val[threadIdx.x] = vals[threadIdx.x];
    if (threadIdx.x == 0) {
        a.ctr->p2p_epoch = next_epoch(epoch);
        const unsigned long long dt_x = (unsigned long long)(wall_clock64() - t1);
        x.p2p_ticks += dt_x;
        x.count += 1ull;
        *xs = x;
        xchg_dist_note(a.ctr, a.xchg_class, dt_x, 0ull);  // (every timed exchange is in a bin of the histogram)
    }
}

// ------------------------------------------------------------------------------------ start-up self-test of the hand-off
// Every workgroup contributes blockIdx + 1 (and its square) through the tagged-granule hand-off the hot kernels use; the
// totals are known exactly.  Run once per context, right after its buffers exist (lbfgs_hip_ctx_create): the granules live
// in pooled UNCACHED memory because recycled uncached blocks were once seen to serve stale data
// (profiles/r03_uncached_granules_recycling.md) -- should a runtime or driver bring that back in another form, the
// context notices here, says so, and takes plain device memory for its granules instead of computing wrong sums.
__global__ __launch_bounds__(BLOCK) void handoff_selftest_kernel(const RedCtl red) {
    const DevCounters c0 = load_counters(red);
    double acc[2] = {0.0, 0.0};
    if (threadIdx.x == 0) {
        acc[0] = (double)(blockIdx.x + 1u);
        acc[1] = acc[0] * acc[0];
    }
    grid_reduce<2>(acc, red, c0);
}

// ------------------------------------------------------------------------------------ plumbing shared with the other units (ctx.h)
int fail(lbfgs_hip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

int grid_for(const lbfgs_hip_ctx* ctx, int x32) {
    int g = ctx->grid_override > 0 ? ctx->grid_override : (x32 == 27 ? ctx->grid_default : std::max(1, ctx->cu_count * x32 / 32));
    if (g > MAX_GRID) g = MAX_GRID;
    if (g < 1) g = 1;
    return g;
}

// the cross-workgroup hand-off part of a RedCtl for the NEXT reducing launch
// `nred`: sums of that launch.  Kernels with more than RED_PTRS sums (the Gram rows) always take the ticket form, which
// neither reads nor advances DevCounters::red_epoch -- the host's shadow of it must not count them either.
int fill_handoff(lbfgs_hip_ctx* ctx, RedCtl& red, int nred) {
    ctx->last_res.valid = false;  // (two_loop_impl sets it again after a resident launch)
    red.partials = ctx->partials;
    red.ticket = ctx->ticket;
    red.gran = ctx->gran;
    red.err = ctx->p2p_err;
    red.timeout_ticks = ctx->handoff_timeout_ticks;  // (10 s of the 100 MHz wall clock unless LBFGS_HIP_HANDOFF_TIMEOUT_MS says otherwise)
    red.xchg_class = (unsigned int)ctx->xchg_class;
    red.ctr = ctx->dev_ctr;
    red.tagged = (ctx->handoff_ticket || nred > RED_PTRS) ? 0u : 1u;
    if (red.tagged) {
        // the device walks the tags 1, 2, ..., 2^32-1, 1, ... by itself; when they come round, no granule of the buffer
        // may still carry a tag from the previous round
        if (ctx->red_count > 0 && ctx->red_count % 0xFFFFFFFFull == 0) {
            HIP_TRY(ctx, hipMemsetAsync(ctx->gran, 0, (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long), ctx->stream));
        }
        ctx->red_count += 1;
    }
    return LBFGS_HIP_OK;
}


// ---- the gated exchange's plumbing (stream.h ext_exchange) ---------------------------------------------------------------
static_assert(P2P_MBOX_WORDS * sizeof(unsigned long long) >= EXT_BUF_OFFSET + (size_t)EXT_SLOTS * EXT_SLOT_DOUBLES * sizeof(double),
              "the gated exchange keeps its flags and slots in one block of the mailbox pool");
// The two streams of the gated exchange must be served CONCURRENTLY: the gate kernel spins while the persistent kernel runs,
// and the persistent kernel waits for what comes behind the gate.  HIP deals its streams onto a handful of hardware queues
// (four by default), so two streams of a process that has many -- PyTorch's, RCCL's -- can share one, and then the second
// kernel never starts while the first waits for it.  The second stream therefore gets the HIGHEST priority (priorities have
// hardware queues of their own), and a handshake at first use proves the concurrency before anything relies on it.
__global__ __launch_bounds__(64) void ext_selftest_wait_kernel(unsigned long long* flags, unsigned long long magic, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    unsigned long long seen = 0;
    while ((seen = __hip_atomic_load(flags + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != magic &&
           (unsigned long long)(wall_clock64() - t0) < ticks)
        __builtin_amdgcn_s_sleep(2);
    __hip_atomic_store(flags + 4, seen == magic ? 1ull : 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void ext_selftest_set_kernel(unsigned long long* flags, unsigned long long magic) {
    if (threadIdx.x == 0) __hip_atomic_store(flags + 3, magic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// *bad (0 / 1: this rank's outcome) -> the number of ranks that reported 1.  The buffer is the board's first private slot (the
// two-loop's ping-pong dots: idle while a context is being created) -- NOT the uncached block, which a rank that failed to
// allocate it does not have and must still vote.
static int ext_agree(lbfgs_hip_ctx* ctx, int* bad) {
    double* const buf = ctx->board + LBFGS_HIP_BOARD_SLOTS;
    double v = *bad ? 1.0 : 0.0;
    HIP_TRY(ctx, hipMemcpy(buf, &v, sizeof(v), hipMemcpyHostToDevice));
    const int rc = rccl_allreduce_on(ctx, buf, 1, ctx->stream);
    if (rc != LBFGS_HIP_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(&v, buf, sizeof(v), hipMemcpyDeviceToHost));
    *bad = (int)v;
    v = 0.0;
    HIP_TRY(ctx, hipMemcpy(buf, &v, sizeof(v), hipMemcpyHostToDevice));
    return LBFGS_HIP_OK;
}
// The steps of ext_prepare as the collective skeleton sees them (ext_protocol.h: every rank makes the same sequence of
// collectives whatever happens to it locally -- a failure is RECORDED, ctx->err keeps the message, and handed back after the last
// one: round-5 advice).
#define EXT_LOCAL(call)                                                                                    \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return fail(ctx, LBFGS_HIP_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
namespace {
struct ExtOps {
    lbfgs_hip_ctx* ctx;
    bool has_comm() const { return ctx->nccl != nullptr; }
    int setup() {
        if (!ctx->xstream) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // (numerically lowest = highest priority)
            EXT_LOCAL(hipStreamCreateWithPriority(&ctx->xstream, hipStreamNonBlocking, hi));
            if (ctx->resident_fault == -2) return fail(ctx, LBFGS_HIP_ERR_HIP, "injected: the second stream could not be created (tests)");
        }
        if (!ctx->ext_ev) EXT_LOCAL(hipEventCreateWithFlags(&ctx->ext_ev, hipEventDisableTiming));
        if (!ctx->ext_block) {
            void* p = nullptr;
            {
                std::lock_guard<std::mutex> lk(g_uc_pool_mu);  // (uncached blocks are pooled per process, never freed: see lbfgs_hip_ctx_create)
                auto& pool = g_uc_mbox_pool[ctx->device];
                if (!pool.empty()) {
                    p = pool.back();
                    pool.pop_back();
                }
            }
            const size_t bytes = P2P_MBOX_WORDS * sizeof(unsigned long long);
            if (!p) EXT_LOCAL(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
            ctx->ext_block = static_cast<unsigned long long*>(p);
            EXT_LOCAL(hipMemsetAsync(p, 0, bytes, ctx->stream));
            EXT_LOCAL(hipStreamSynchronize(ctx->stream));
        }
        return LBFGS_HIP_OK;
    }
    // a kernel on the compute stream waits (<= 20 ms) for a word that a kernel on the second stream sets
    int handshake(bool* passed) {
        const unsigned long long magic = 0x6761746564ull;
        hipLaunchKernelGGL(ext_selftest_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->ext_block, magic, 2000000ull);
        hipLaunchKernelGGL(ext_selftest_set_kernel, dim3(1), dim3(64), 0, ctx->xstream, ctx->ext_block, magic);
        EXT_LOCAL(hipGetLastError());
        EXT_LOCAL(hipStreamSynchronize(ctx->stream));
        EXT_LOCAL(hipStreamSynchronize(ctx->xstream));
        unsigned long long verdict = 0;
        EXT_LOCAL(hipMemcpy(&verdict, ctx->ext_block + 4, sizeof(verdict), hipMemcpyDeviceToHost));
        *passed = verdict == 1ull;
        return LBFGS_HIP_OK;
    }
    // an empty shard, a shard beyond the kernel's limits, a CU-masked queue cannot take the persistent kernel (lbfgs_hip.hip)
    bool shard_ok() { return resident_shard_eligible(ctx); }
    int agree(int* bad) { return ext_agree(ctx, bad); }
    // RCCL's first collective on a stream may set things up (channels, buffers) and wait for the device: let it do so NOW, not
    // with a persistent kernel waiting for it
    int warmup() { return rccl_allreduce_on(ctx, slot0(), 1, ctx->xstream); }
    int after_warmup() {
        EXT_LOCAL(hipStreamSynchronize(ctx->xstream));
        EXT_LOCAL(hipMemsetAsync(slot0(), 0, EXT_SLOT_DOUBLES * sizeof(double), ctx->stream));
        EXT_LOCAL(hipStreamSynchronize(ctx->stream));
        return LBFGS_HIP_OK;
    }
    // the exchange itself, once, under a kernel that fills the chip the way the two-loop will
    int trial(bool* ok, bool* enqueued) {
        if (ctx->resident_fault == -3) return fail(ctx, LBFGS_HIP_ERR_HIP, "injected: this rank failed before the gated exchange's trial (tests)");
        const int rc = ext_selftest_resident(ctx, ok, enqueued);
        if (ctx->resident_fault == -1) *ok = false;  // (tests)
        return rc;
    }
    int bare_allreduce() {  // (on the compute stream, ungated)
        const int rc = rccl_allreduce_on(ctx, ctx->board + LBFGS_HIP_BOARD_SLOTS, 1, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    double* slot0() const { return reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->ext_block) + EXT_BUF_OFFSET); }
};
}  // namespace
#undef EXT_LOCAL
// -> LBFGS_HIP_OK: the gated exchange is usable on EVERY rank; 1: not (every rank takes the kernel-per-step form); < 0: error.
// COLLECTIVE over the context's communicator.
int ext_prepare(lbfgs_hip_ctx* ctx) {
    ExtOps ops{ctx};
    ExtOutcome res;
    const int rc = ext_prepare_protocol(ops, &res);
    if (rc != 1) return rc;
    ctx->rccl_resident = 0;
    if (res.stage == 0)
        fprintf(stderr, "[lbfgs_hip] note: the shard or the device of %d of %d ranks cannot take the persistent two-loop kernel (an empty "
                        "shard, a shard beyond its limits, a CU-masked queue); every rank runs the two-loop with a kernel per step "
                        "under RCCL\n", res.bad, ctx->shard.world);
    else if (res.stage == 1)
        fprintf(stderr, "[lbfgs_hip] warning: the gated RCCL exchange is not available on %d of %d ranks (the compute stream and the "
                        "second stream are not served concurrently -- they share a hardware queue: GPU_MAX_HW_QUEUES --, or a "
                        "shard / device cannot take the persistent kernel); the two-loop runs with a kernel per step under RCCL\n",
                res.bad, ctx->shard.world);
    else
        fprintf(stderr, "[lbfgs_hip] warning: the gated RCCL exchange's self-test (an all-reduce enqueued on the second stream, "
                        "awaited by a kernel that fills the chip) did not come back right within 1 s on %d of %d ranks; the "
                        "two-loop runs with a kernel per step under RCCL\n", res.bad, ctx->shard.world);
    return 1;  // (not an error: the caller takes the kernel-per-step form)
}
// Give the chain of the latest gated launch up: its gates leave at once (abort word >= their launch id), the all-reduces behind
// them reduce whatever the slots hold -- on every rank alike, nobody reads it -- and the second stream runs empty.
void ext_abort(lbfgs_hip_ctx* ctx) {
    if (!ctx->xstream || !ctx->ext_block) return;
    const unsigned long long id = ctx->ext_launches;
    (void)hipMemcpy(ctx->ext_block + 2, &id, sizeof(id), hipMemcpyHostToDevice);
    // The drain is BOUNDED: an all-reduce of the chain that a peer never enqueues (the peer died, or took another path after an
    // error of its own) does not complete, and a hipStreamSynchronize here would hold lbfgs_hip_ctx_destroy and every error
    // return for ever.  Poll; past the deadline abort the communicator (ncclCommAbort ends its kernels), poll once more.
    auto drained = [&](double seconds) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(ctx->xstream);
            if (q != hipErrorNotReady) {
                (void)hipGetLastError();
                return true;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    };
    const double wait_s = std::max(2.0, 2.0 * (double)(ctx->p2p_timeout_ticks + ctx->handoff_timeout_ticks) * 1e-8);
    if (drained(wait_s)) return;
    fprintf(stderr, "[lbfgs_hip] warning: the gated RCCL chain did not drain within %.0f s (a peer never enqueued its all-reduce?)%s\n",
            wait_s, (ctx->nccl && g_rccl.CommAbort) ? ": aborting the communicator" : "");
    if (ctx->nccl && g_rccl.CommAbort) {
        (void)g_rccl.CommAbort(ctx->nccl);
        ctx->nccl = nullptr;  // (aborted communicators are freed by the abort)
        (void)drained(2.0);
    }
}
int rccl_allreduce_on(lbfgs_hip_ctx* ctx, double* buf, int count, hipStream_t stream) {
    if (!ctx->nccl) return fail(ctx, LBFGS_HIP_ERR_COMM, "the RCCL communicator of this context was aborted (a gated chain did not drain)");
    const int rc = g_rccl.AllReduce(buf, buf, (size_t)count, kNcclDouble, kNcclSum, ctx->nccl, stream);
    if (rc != 0) return fail(ctx, LBFGS_HIP_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString(rc));
    return LBFGS_HIP_OK;
}

P2PCtl next_p2p(lbfgs_hip_ctx* ctx) {
    P2PCtl c{};
    for (int r = 0; r < ctx->shard.world && r < P2P_MAX_WORLD; ++r) c.mbox[r] = ctx->p2p_mbox[r];
    c.world = ctx->shard.world;
    c.rank = ctx->shard.rank;
    ctx->p2p_count += 1;
    if (ctx->xchg_class) ctx->two_loop_exchanges += 1;
    c.err = ctx->p2p_err;
    c.timeout_ticks = ctx->p2p_timeout_ticks;
    return c;
}

// ---- closing a reduction across ranks --------------------------------------------------------
// `ptrs` are device addresses (board or history scalars) just written by the last workgroup.
int allreduce(lbfgs_hip_ctx* ctx, double* const* ptrs, int count) {
    if (ctx->comm_kind == LBFGS_HIP_COMM_NONE || count == 0) return LBFGS_HIP_OK;
    for (int i = 0; i < count; ++i) {  // the reduced values will not be in the host mirror
        const long idx = ptrs[i] - ctx->board;
        if (idx >= 0 && idx < LBFGS_HIP_BOARD_SLOTS + 2) ctx->mirror_valid[idx] = false;
    }
    ProfScope ps(ctx, LBFGS_HIP_K_COMM);
    ctx->allreduce_calls += 1;
    if (ctx->xchg_class && ctx->comm_kind != LBFGS_HIP_COMM_P2P) ctx->two_loop_exchanges += 1;  // (P2P: next_p2p counts)
    if (ctx->comm_kind == LBFGS_HIP_COMM_RCCL) {
        if (!ctx->nccl) return fail(ctx, LBFGS_HIP_ERR_COMM, "the RCCL communicator of this context was aborted (a gated chain did not drain)");
        // coalesce runs of consecutive addresses into one message each; group them into one launch
        int rc = g_rccl.GroupStart();
        if (rc != 0) return fail(ctx, LBFGS_HIP_ERR_COMM, "ncclGroupStart: %s", g_rccl.GetErrorString(rc));
        int i = 0;
        while (i < count) {
            int j = i + 1;
            while (j < count && ptrs[j] == ptrs[j - 1] + 1) ++j;
            rc = g_rccl.AllReduce(ptrs[i], ptrs[i], (size_t)(j - i), kNcclDouble, kNcclSum, ctx->nccl, ctx->stream);
            if (rc != 0) {
                g_rccl.GroupEnd();
                return fail(ctx, LBFGS_HIP_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString(rc));
            }
            i = j;
        }
        rc = g_rccl.GroupEnd();
        if (rc != 0) return fail(ctx, LBFGS_HIP_ERR_COMM, "ncclGroupEnd: %s", g_rccl.GetErrorString(rc));
        return LBFGS_HIP_OK;
    }
    if (ctx->comm_kind == LBFGS_HIP_COMM_P2P) {
        P2PArgs a{};
        a.ctl = next_p2p(ctx);
        a.ctr = ctx->dev_ctr;
        for (int i = 0; i < count; ++i) a.val[i] = ptrs[i];
        a.count = count;
        a.xchg_class = (unsigned int)ctx->xchg_class;
        hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(64), 0, ctx->stream, a);
        HIP_TRY(ctx, hipGetLastError());
        return LBFGS_HIP_OK;
    }
    // host callback: stage through pinned memory (synchronises the stream)
    for (int i = 0; i < count; ++i)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned + i, ptrs[i], sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->cb(ctx->cb_user, ctx->pinned, count) != 0)
        return fail(ctx, LBFGS_HIP_ERR_COMM, "all-reduce callback failed");
    for (int i = 0; i < count; ++i)
        HIP_TRY(ctx, hipMemcpyAsync(ptrs[i], ctx->pinned + i, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    // the staging buffer is reused by the next call: make sure the uploads have left it
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LBFGS_HIP_OK;
}

// ---- the reduction-control block of the NEXT launch with NRED sums (hand-off, P2P exchange, host mirror) ------
int prep_red(lbfgs_hip_ctx* ctx, RedCtl& red, int nred, double* const* red_out, double* dup_ptr, int dup_k,
             bool* in_kernel_exchange) {
    red.dup_ptr = dup_ptr;
    red.dup_k = dup_k;
    if (nred > 0) {
        const int rc_h = fill_handoff(ctx, red, nred);
        if (rc_h != LBFGS_HIP_OK) return rc_h;
    }
    if (nred <= RED_PTRS) {
        for (int k = 0; k < nred; ++k) red.out[k] = red_out[k];
    } else {
        red.out_contig = red_out[0];  // caller guarantees red_out[k] == red_out[0] + k
    }
    *in_kernel_exchange = nred > 0 && ctx->comm_kind == LBFGS_HIP_COMM_P2P;
    if (*in_kernel_exchange) red.p2p = next_p2p(ctx);  // the last workgroup closes the reduction itself
    if (nred > 0 && nred <= RED_PTRS) {
        // totals are final inside the kernel (one rank, or in-kernel exchange): mirror them to the host -- unless every
        // destination is one of the private slots past the public board (the two-loop's inner dot products), which
        // the host can never ask for: those kernels skip the two PCIe stores and the sequence word
        const bool final_in_kernel = ctx->comm_kind == LBFGS_HIP_COMM_NONE || *in_kernel_exchange;
        bool any_public = false;
        for (int k = 0; k < nred; ++k) {
            const long idx = red_out[k] - ctx->board;
            if (idx >= 0 && idx < LBFGS_HIP_BOARD_SLOTS) any_public = true;
        }
        const bool mirrored = final_in_kernel && ctx->mirror && any_public;
        for (int k = 0; k < nred; ++k) {
            const long idx = red_out[k] - ctx->board;
            if (idx >= 0 && idx < LBFGS_HIP_BOARD_SLOTS + 2) {
                ctx->mirror_valid[idx] = mirrored;
            }
        }
        if (mirrored) {
            red.mirror.host_board = ctx->mirror_dev;
            red.mirror.board = ctx->board;
            red.mirror.host_seq = reinterpret_cast<unsigned long long*>(ctx->mirror_dev + LBFGS_HIP_BOARD_SLOTS + 2);
            ctx->mirror_seq += 1;  // (the device counts the same way: DevCounters::mirror_seq)
            red.mirror.host_err = reinterpret_cast<unsigned long long*>(ctx->mirror_dev + LBFGS_HIP_BOARD_SLOTS + 3);
            red.mirror.slots = LBFGS_HIP_BOARD_SLOTS + 2;
        }
    }
    return LBFGS_HIP_OK;
}

namespace {
int prof_drain(lbfgs_hip_ctx* ctx) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& pc : ctx->prof) {
        for (auto& pr : pc.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
                pc.ms += ms;
                pc.launches += 1;
            }
            ctx->prof_pool.push_back(pr);
        }
        pc.pending.clear();
    }
    return LBFGS_HIP_OK;
}

}  // namespace

// ==================================================================================== context
extern "C" {

int lbfgs_hip_abi_version(void) { return LBFGS_HIP_ABI_VERSION; }
const char* lbfgs_hip_build_id(void) { return lbfgs_hip_build_id_marker + sizeof("LBFGS_HIP_BUILD_ID=") - 1; }

int lbfgs_hip_device_count(int* count) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, LBFGS_HIP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = c;
    return LBFGS_HIP_OK;
}

int lbfgs_hip_rccl_unique_id(void* out128) {
    std::string err;
    if (!out128) return fail(nullptr, LBFGS_HIP_ERR_ARG, "null id buffer");
    if (!rccl_load(&err)) return fail(nullptr, LBFGS_HIP_ERR_COMM, "%s", err.c_str());
    nccl_unique_id_t id;
    int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return fail(nullptr, LBFGS_HIP_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString(rc));
    memcpy(out128, &id, sizeof(id));
    return LBFGS_HIP_OK;
}

int lbfgs_hip_p2p_mailbox_create(int device, void** mailbox_out, void* ipc_handle64_out) {
    return lbfgs_hip_p2p_mailbox_create2(device, LBFGS_HIP_MAILBOX_AUTO, mailbox_out, ipc_handle64_out);
}

int lbfgs_hip_p2p_mailbox_create2(int device, int placement, void** mailbox_out, void* ipc_handle64_out) {
    if (!mailbox_out || !ipc_handle64_out) return fail(nullptr, LBFGS_HIP_ERR_ARG, "null argument");
    *mailbox_out = nullptr;
    if (placement == LBFGS_HIP_MAILBOX_AUTO) {
        const char* e = getenv("LBFGS_HIP_P2P_MAILBOX");
        placement = (e && strcmp(e, "host") == 0) ? LBFGS_HIP_MAILBOX_HOST : LBFGS_HIP_MAILBOX_DEVICE;
    }
    if (placement != LBFGS_HIP_MAILBOX_DEVICE && placement != LBFGS_HIP_MAILBOX_HOST)
        return fail(nullptr, LBFGS_HIP_ERR_ARG, "unknown mailbox placement %d", placement);
    hipError_t e = hipSetDevice(device);
    const size_t bytes = P2P_MBOX_WORDS * sizeof(unsigned long long);
    if (placement == LBFGS_HIP_MAILBOX_HOST) {
        if (e != hipSuccess) return fail(nullptr, LBFGS_HIP_ERR_COMM, "P2P mailbox: %s", hipGetErrorString(e));
        static std::atomic<unsigned int> serial{0};
        char name[48];
        snprintf(name, sizeof(name), "/lbfgs_hip_mbox_%ld_%u_%08x", (long)getpid(), serial.fetch_add(1), (unsigned int)std::chrono::steady_clock::now().time_since_epoch().count());
        std::string err;
        void* dev = host_mbox_map(name, bytes, true, &err);
        if (!dev) return fail(nullptr, LBFGS_HIP_ERR_COMM, "P2P mailbox (host placement): %s", err.c_str());
        memset(ipc_handle64_out, 0, HIP_IPC_HANDLE_SIZE);
        memcpy(ipc_handle64_out, HOST_MBOX_MAGIC, sizeof(HOST_MBOX_MAGIC));
        memcpy((char*)ipc_handle64_out + sizeof(HOST_MBOX_MAGIC), name, strlen(name) + 1);
        *mailbox_out = dev;
        return LBFGS_HIP_OK;
    }
    void* p = nullptr;
    if (e == hipSuccess) {  // (a retired mailbox of this process first: uncached blocks are pooled, never freed)
        std::lock_guard<std::mutex> lk(g_uc_pool_mu);
        auto& pool = g_uc_mbox_pool[device];
        if (!pool.empty()) {
            p = pool.back();
            pool.pop_back();
        }
    }
    if (e == hipSuccess && !p) e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(p, 0, bytes);  // tag 0 is never a valid epoch
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t hdl;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&hdl, p);
    if (e != hipSuccess) {
        if (p) uc_mbox_retire(device, p);
        return fail(nullptr, LBFGS_HIP_ERR_COMM, "P2P mailbox: %s", hipGetErrorString(e));
    }
    local_mbox_register(hdl, p, device);
    memcpy(ipc_handle64_out, &hdl, HIP_IPC_HANDLE_SIZE);
    *mailbox_out = p;
    return LBFGS_HIP_OK;
}

void lbfgs_hip_p2p_mailbox_destroy(int device, void* mailbox) {
    if (!mailbox) return;
    (void)hipSetDevice(device);
    if (host_mbox_release(mailbox)) return;
    (void)hipDeviceSynchronize();
    uc_mbox_retire(device, mailbox);
}

int lbfgs_hip_ctx_create(lbfgs_hip_ctx** out, int device, uint64_t n, const lbfgs_hip_shard* shard,
                         const lbfgs_hip_comm* comm, void* stream) {
    if (!out) return fail(nullptr, LBFGS_HIP_ERR_ARG, "null out");
    *out = nullptr;
    // A P2P mailbox handed in through `comm` belongs to this call from here on, whether it succeeds or not
    // (lbfgs_hip.h): until the context owns it, every failure path frees it.
    void* const mbox_in = (comm && comm->kind == LBFGS_HIP_COMM_P2P) ? comm->p2p_mailbox : nullptr;
    auto drop_mbox = [&]() {
        if (mbox_in) lbfgs_hip_p2p_mailbox_destroy(device, mbox_in);
    };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        drop_mbox();
        return fail(nullptr, LBFGS_HIP_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    }
    if (device < 0 || device >= ndev) {
        drop_mbox();
        return fail(nullptr, LBFGS_HIP_ERR_ARG, "device %d out of range (%d)", device, ndev);
    }

    lbfgs_hip_ctx* ctx = new (std::nothrow) lbfgs_hip_ctx();
    if (!ctx) {
        drop_mbox();
        return fail(nullptr, LBFGS_HIP_ERR_NOMEM, "out of host memory");
    }
    ctx->device = device;
    if (shard) {
        ctx->shard = *shard;
        if (shard->world < 1 || shard->rank < 0 || shard->rank >= shard->world ||
            shard->offset + shard->n_local > shard->n_global) {
            delete ctx;
            drop_mbox();
            return fail(nullptr, LBFGS_HIP_ERR_ARG, "inconsistent shard");
        }
    } else {
        ctx->shard.rank = 0;
        ctx->shard.world = 1;
        ctx->shard.n_global = n;
        ctx->shard.offset = 0;
        ctx->shard.n_local = n;
    }
    if (mbox_in) {
        if (ctx->shard.rank >= P2P_MAX_WORLD) {
            delete ctx;
            drop_mbox();
            return fail(nullptr, LBFGS_HIP_ERR_ARG, "P2P communicator supports world <= %d", P2P_MAX_WORLD);
        }
        ctx->p2p_mbox[ctx->shard.rank] = (unsigned long long*)mbox_in;  // lbfgs_hip_ctx_destroy frees it from now on
    }
#define CTX_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            int rc_ = fail(nullptr, LBFGS_HIP_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));   \
            lbfgs_hip_ctx_destroy(ctx);                                                            \
            return rc_;                                                                            \
        }                                                                                          \
    } while (0)
    CTX_TRY(hipSetDevice(device));
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        CTX_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    hipDeviceProp_t prop;
    CTX_TRY(hipGetDeviceProperties(&prop, device));
    // ~0.85 workgroups per CU: measured optimum of the windowed streaming map on MI355X (216 of 256 CUs);
    // more workgroups only add concurrent DRAM pages (tools/tune_stream.hip)
    ctx->cu_count = prop.multiProcessorCount;
    ctx->grid_default = std::max(1, prop.multiProcessorCount * 27 / 32);
    ctx->gram_grid = prop.multiProcessorCount;  // the 21-stream Gram rows pass peaks at one workgroup per CU
    if (const char* e = getenv("LBFGS_HIP_NT_THRESHOLD_MB")) ctx->nt_threshold_bytes = (size_t)atoll(e) << 20;
    if (const char* e = getenv("LBFGS_HIP_DEFER_SUMS")) ctx->defer_inner_sums = atoi(e) != 0;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT")) ctx->resident_on = atoi(e) != 0;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_GRID")) ctx->resident_grid = std::max(0, atoi(e));
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_FAULT")) ctx->resident_fault = atoi(e);
    if (const char* e = getenv("LBFGS_HIP_GRAM_COMBINE_RESIDENT")) ctx->gram_combine_resident = atoi(e) != 0;
    if (const char* e = getenv("LBFGS_HIP_LJ_BUILD_FP32")) ctx->lj_build_fp32 = atoi(e) != 0;
    if (const char* e = getenv("LBFGS_HIP_LJ_FUSED_TRIAL")) ctx->lj_fused_trial = atoi(e) != 0;
    if (const char* e = getenv("LBFGS_HIP_RCCL_RESIDENT")) ctx->rccl_resident = atoi(e) != 0;
    if (const char* e = getenv("LBFGS_HIP_HANDOFF_TIMEOUT_MS")) ctx->handoff_timeout_ticks = (unsigned long long)std::max(1, atoi(e)) * 100000ULL;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_FIRST_TIMEOUT_MS")) ctx->first_timeout_ticks = (unsigned long long)std::max(1, atoi(e)) * 100000ULL;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_NT_MB")) ctx->resident_nt_bytes = (size_t)std::max(0, atoi(e)) << 20;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_HYBRID")) ctx->resident_hybrid = atoi(e) != 0;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_TOUCH")) ctx->resident_touch = std::min(64, std::max(-1, atoi(e)));  // (the kernel caps it at what it was built with)
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_PLAIN_MB")) ctx->resident_plain_bytes = (size_t)std::max(0, atoi(e)) << 20;
    if (const char* e = getenv("LBFGS_HIP_NT_STORE_THRESHOLD_MB")) ctx->nt_store_threshold_bytes = (size_t)atoll(e) << 20;
    if (const char* e = getenv("LBFGS_HIP_HANDOFF")) ctx->handoff_ticket = strcmp(e, "ticket") == 0;
    if (const char* e = getenv("LBFGS_HIP_GRID")) ctx->grid_override = std::min(MAX_GRID, std::max(0, atoi(e)));
    for (int k = 0; k < LBFGS_HIP_K_CLASSES; ++k) {  // tuning experiments: the launch grid of one kernel class
        char name[40];
        snprintf(name, sizeof(name), "LBFGS_HIP_GRID_X32_K%d", k);
        if (const char* e = getenv(name)) ctx->grid_x32_class[k] = std::min(512, std::max(0, atoi(e)));
    }
    CTX_TRY(hipMalloc(&ctx->board, (LBFGS_HIP_BOARD_SLOTS + 2) * sizeof(double)));
    CTX_TRY(hipMemsetAsync(ctx->board, 0, (LBFGS_HIP_BOARD_SLOTS + 2) * sizeof(double), ctx->stream));
    CTX_TRY(hipMalloc(&ctx->partials, (size_t)MAX_RED * MAX_GRID * sizeof(double)));
    CTX_TRY(hipMalloc(&ctx->ticket, 64));
    CTX_TRY(hipMemsetAsync(ctx->ticket, 0, 64, ctx->stream));
    {
        // The tagged granules live in UNCACHED device memory: every access to them is an agent-scope atomic that must be
        // coherent across the eight XCDs' private L2s anyway, and on memory the L2s never hold, such an access takes a shorter
        // path -- a chip-wide hand-off of one sum among 256 workgroups: 2.0 us instead of 3.0 (tools/handoff_bench.hip,
        // profiles/r03_handoff_bench.log).  LBFGS_HIP_GRAN_CACHED=1: plain hipMalloc, for A/B.
        const size_t gbytes = (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long);
        void* gp = nullptr;
        // Uncached blocks are POOLED per process and device and never handed back to the allocator: memory that changes
        // between cached and uncached use as contexts come and go was seen to go wrong (a context created right after
        // another one had been destroyed occasionally read stale data in its first reductions -- always with vectors of
        // 2-5 MB, the size of this block -- as long as this block was allocated and freed per context; never once it
        // stopped being recycled: profiles/r03_uncached_granules_recycling.md).
        const char* gc = getenv("LBFGS_HIP_GRAN_CACHED");
        if (!(gc && *gc && atoi(gc) != 0)) {
            std::lock_guard<std::mutex> lk(g_uc_pool_mu);
            auto& pool = g_uc_pool[device];
            if (!pool.empty()) {
                gp = pool.back();
                pool.pop_back();
            } else if (hipExtMallocWithFlags(&gp, gbytes, hipDeviceMallocUncached) != hipSuccess) {
                (void)hipGetLastError();
                gp = nullptr;
            }
            ctx->gran_pooled = gp != nullptr;
        }
        if (!gp) CTX_TRY(hipMalloc(&gp, gbytes));
        ctx->gran = (unsigned long long*)gp;
    }
    CTX_TRY(hipMemsetAsync(ctx->gran, 0, (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long), ctx->stream));
    {
        // The first tag of a context is not 1 but a value no other context of this process starts near: a granule buffer is
        // recycled memory, and although it is zeroed above, a poller that found a stale granule of an EARLIER context with
        // the tag it is waiting for would take it for a partial sum (seen once the buffer moved to uncached memory: the same
        // case failed in every full run of the GPU suite and never alone).  Tags are unique per launch within a context
        // (stream.h DevCounters); this makes them unique across the contexts that may have owned the memory before.
        static std::atomic<unsigned int> serial{0};
        const unsigned int epoch0 = 1u + (unsigned int)(((unsigned long long)(serial.fetch_add(1) + 1u) * 0x9E3779B1ull) % 0xFFFFFFFEull);
        ctx->red_count = (unsigned long long)epoch0 - 1ull;  // (the next tagged launch uses tag red_count % (2^32-1) + 1 = epoch0)
        const DevCounters init{epoch0, 1u, 0ull};
        CTX_TRY(hipMalloc(&ctx->dev_ctr, 4096));  // (a page of its own: nothing else shares its cache lines)
        CTX_TRY(hipMemset(ctx->dev_ctr, 0, 4096));  // (the DevXchg records behind the counters start at zero)
        CTX_TRY(hipMemcpy(ctx->dev_ctr, &init, sizeof(init), hipMemcpyHostToDevice));
    }
    CTX_TRY(hipMalloc(&ctx->dot_parts, 2 * (size_t)MAX_GRID * sizeof(double)));
    CTX_TRY(hipMalloc(&ctx->p2p_err, 64));  // device error flag: 1 = a P2P peer never arrived, 2 = a partial never arrived
    CTX_TRY(hipMemsetAsync(ctx->p2p_err, 0, 64, ctx->stream));
    CTX_TRY(hipHostMalloc(&ctx->pinned, (LBFGS_HIP_BOARD_SLOTS + 1) * sizeof(double), hipHostMallocDefault));
    if (!getenv("LBFGS_HIP_NO_MIRROR")) {
        void* hm = nullptr;
        if (hipHostMalloc(&hm, (LBFGS_HIP_BOARD_SLOTS + 4) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) ==
            hipSuccess) {
            memset(hm, 0, (LBFGS_HIP_BOARD_SLOTS + 4) * sizeof(double));
            void* dv = nullptr;
            if (hipHostGetDevicePointer(&dv, hm, 0) == hipSuccess) {
                ctx->mirror = (double*)hm;
                ctx->mirror_dev = (double*)dv;
            } else {
                (void)hipHostFree(hm);
            }
        }
    }
    CTX_TRY(hipStreamSynchronize(ctx->stream));
    // ---- known-answer reductions through the hand-off before anything trusts it (see handoff_selftest_kernel); two rounds:
    // consecutive tags, both granule halves.  A mismatch with pooled uncached granules -> plain device memory, once.
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool ok = true;
        const int G = std::max(1, std::min(ctx->cu_count, MAX_GRID));
        for (int round = 0; round < 2 && ok; ++round) {
            RedCtl red{};
            if (fill_handoff(ctx, red, 2) != LBFGS_HIP_OK) { ok = false; break; }
            red.out[0] = ctx->board + LBFGS_HIP_BOARD_SLOTS;      // (the two private slots past the public board)
            red.out[1] = ctx->board + LBFGS_HIP_BOARD_SLOTS + 1;
            hipLaunchKernelGGL(handoff_selftest_kernel, dim3(G), dim3(BLOCK), 0, ctx->stream, red);
            double got[2] = {0.0, 0.0};
            CTX_TRY(hipMemcpyAsync(got, red.out[0], sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
            CTX_TRY(hipStreamSynchronize(ctx->stream));
            const double g = (double)G;
            ok = got[0] == 0.5 * g * (g + 1.0) && got[1] == g * (g + 1.0) * (2.0 * g + 1.0) / 6.0;
        }
        if (ok) break;
        if (attempt == 1 || !ctx->gran_pooled) {
            lbfgs_hip_ctx_destroy(ctx);
            return fail(nullptr, LBFGS_HIP_ERR_HIP, "the cross-workgroup hand-off failed its start-up self-test (known-answer reduction)");
        }
        int rt = 0, drv = 0;
        (void)hipRuntimeGetVersion(&rt);
        (void)hipDriverGetVersion(&drv);
        fprintf(stderr, "[lbfgs_hip] warning: the hand-off's start-up self-test failed with granules in uncached device memory (HIP "
                        "runtime %d, driver %d); this context keeps them in plain device memory instead (LBFGS_HIP_GRAN_CACHED=1 "
                        "selects that from the start)\n", rt, drv);
        {   // the suspect block goes back to the pool (never to the allocator: see above); a plain one takes its place
            std::lock_guard<std::mutex> lk(g_uc_pool_mu);
            g_uc_pool[device].push_back(ctx->gran);
        }
        ctx->gran = nullptr;
        ctx->gran_pooled = false;
        void* gp = nullptr;
        CTX_TRY(hipMalloc(&gp, (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long)));
        ctx->gran = (unsigned long long*)gp;
        CTX_TRY(hipMemsetAsync(ctx->gran, 0, (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long), ctx->stream));
        CTX_TRY(hipMemsetAsync(ctx->p2p_err, 0, 64, ctx->stream));
        CTX_TRY(hipStreamSynchronize(ctx->stream));
    }
#undef CTX_TRY

    const int kind = comm ? comm->kind : LBFGS_HIP_COMM_NONE;
    if (ctx->shard.world > 1 && kind == LBFGS_HIP_COMM_NONE) {
        lbfgs_hip_ctx_destroy(ctx);
        return fail(nullptr, LBFGS_HIP_ERR_ARG, "world > 1 needs a communicator");
    }
    if (kind == LBFGS_HIP_COMM_RCCL) {
        std::string err;
        if (!comm->rccl_unique_id || !rccl_load(&err)) {
            lbfgs_hip_ctx_destroy(ctx);
            return fail(nullptr, LBFGS_HIP_ERR_COMM, "RCCL unavailable: %s", err.c_str());
        }
        nccl_unique_id_t id;
        memcpy(&id, comm->rccl_unique_id, sizeof(id));
        int rc = g_rccl.CommInitRank(&ctx->nccl, ctx->shard.world, id, ctx->shard.rank);
        if (rc != 0) {
            int r = fail(nullptr, LBFGS_HIP_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(rc));
            ctx->nccl = nullptr;
            lbfgs_hip_ctx_destroy(ctx);
            return r;
        }
        // what RCCL itself reports must be the shard this context was given (a communicator that spans fewer ranks than the
        // vector is sharded over would silently produce partial sums)
        int cnt = 0, urank = -1;
        const int rc1 = g_rccl.CommCount(ctx->nccl, &cnt), rc2 = g_rccl.CommUserRank(ctx->nccl, &urank);
        if (rc1 != 0 || rc2 != 0 || cnt != ctx->shard.world || urank != ctx->shard.rank) {
            int r = fail(nullptr, LBFGS_HIP_ERR_COMM, "the RCCL communicator reports %d ranks / rank %d (status %d, %d), the shard says %d / %d",
                         cnt, urank, rc1, rc2, ctx->shard.world, ctx->shard.rank);
            lbfgs_hip_ctx_destroy(ctx);
            return r;
        }
        ctx->rccl_ranks_seen = cnt;
        ctx->rccl_rank_seen = urank;
        ctx->comm_kind = LBFGS_HIP_COMM_RCCL;
        ctx->p2p_exclusive = comm->exclusive_device != 0;  // (the persistent kernel with the gated exchange needs the GPU to itself)
        const double tmo = comm->p2p_timeout_s > 0 ? comm->p2p_timeout_s : 5.0;
        ctx->p2p_timeout_ticks = (unsigned long long)(tmo * 1e8);
        if (ctx->p2p_exclusive && ctx->rccl_resident) {  // the gated exchange's second stream, slots, handshake and RCCL warm-up
            const int rc_x = ext_prepare(ctx);
            if (rc_x < 0) {
                const std::string msg = ctx->err;
                lbfgs_hip_ctx_destroy(ctx);
                return fail(nullptr, rc_x, "%s", msg.c_str());
            }
        }
    } else if (kind == LBFGS_HIP_COMM_P2P) {
        const int W = ctx->shard.world, me = ctx->shard.rank;
        if (W > P2P_MAX_WORLD || !comm->p2p_mailbox || !comm->p2p_handles) {
            lbfgs_hip_ctx_destroy(ctx);
            return fail(nullptr, LBFGS_HIP_ERR_ARG, "P2P communicator needs a mailbox, %d handles and world <= %d", W,
                        P2P_MAX_WORLD);
        }
        for (int r = 0; r < W; ++r) {  // (p2p_mbox[me] was taken over at the top)
            if (r == me) continue;
            const char* hraw = (const char*)comm->p2p_handles + (size_t)r * HIP_IPC_HANDLE_SIZE;
            void* p = nullptr;
            if (memcmp(hraw, HOST_MBOX_MAGIC, sizeof(HOST_MBOX_MAGIC)) == 0) {  // HOST placement: the segment's name
                char name[HIP_IPC_HANDLE_SIZE - sizeof(HOST_MBOX_MAGIC) + 1] = {0};
                memcpy(name, hraw + sizeof(HOST_MBOX_MAGIC), HIP_IPC_HANDLE_SIZE - sizeof(HOST_MBOX_MAGIC));
                std::string err;
                p = host_mbox_map(name, P2P_MBOX_WORDS * sizeof(unsigned long long), false, &err);
                if (!p) {
                    int rc = fail(nullptr, LBFGS_HIP_ERR_COMM, "mailbox of rank %d (host placement): %s", r, err.c_str());
                    lbfgs_hip_ctx_destroy(ctx);
                    return rc;
                }
                ctx->p2p_mbox[r] = (unsigned long long*)p;  // (released through the registry: p2p_opened stays false)
                ctx->p2p_peers_host += 1;
                continue;
            }
            hipIpcMemHandle_t hdl;
            memcpy(&hdl, hraw, sizeof(hdl));
            // (LBFGS_HIP_TEST_FAIL_IPC_OPEN=1: tests of the host-placement fallback pretend the mapping is refused)
            const char* fail_hook = getenv("LBFGS_HIP_TEST_FAIL_IPC_OPEN");
            LocalMbox lm{nullptr, -1};
            hipError_t e;
            if (fail_hook && atoi(fail_hook) != 0) {
                e = hipErrorInvalidValue;
            } else if (local_mbox_lookup(hdl, &lm)) {  // a rank hosted by this same process: its mailbox is addressable as it is
                e = hipSuccess;
                p = lm.p;
                if (lm.device != ctx->device) {  // (one process driving several GPUs: the stores need peer access)
                    const hipError_t pe = hipDeviceEnablePeerAccess(lm.device, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) e = pe;
                    (void)hipGetLastError();
                }
            } else {
                e = ipc_open_cached(hdl, &p);
            }
            if (e != hipSuccess) {
                int rc = fail(nullptr, LBFGS_HIP_ERR_COMM, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
                lbfgs_hip_ctx_destroy(ctx);
                return rc;
            }
            ctx->p2p_mbox[r] = (unsigned long long*)p;
            ctx->p2p_opened[r] = true;
            ctx->p2p_peers_device += 1;
        }
        const double tmo = comm->p2p_timeout_s > 0 ? comm->p2p_timeout_s : 5.0;
        ctx->p2p_timeout_ticks = (unsigned long long)(tmo * 1e8);  // wall_clock64 runs at 100 MHz
        ctx->p2p_exclusive = comm->exclusive_device != 0;
        ctx->comm_kind = LBFGS_HIP_COMM_P2P;
    } else if (kind == LBFGS_HIP_COMM_CALLBACK) {
        if (!comm->callback) {
            lbfgs_hip_ctx_destroy(ctx);
            return fail(nullptr, LBFGS_HIP_ERR_ARG, "callback communicator without a callback");
        }
        ctx->cb = comm->callback;
        ctx->cb_user = comm->callback_user;
        ctx->comm_kind = LBFGS_HIP_COMM_CALLBACK;
    }
    *out = ctx;
    return LBFGS_HIP_OK;
}

void lbfgs_hip_ctx_destroy(lbfgs_hip_ctx* ctx) {
    if (!ctx) return;
#if LH_RES_TRACE
    res_trace_print();
#endif
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ext_abort(ctx);  // (a gated chain that is still waiting for a kernel that will never come)
    if (ctx->nccl && g_rccl.ok) g_rccl.CommDestroy(ctx->nccl);
    if (ctx->xstream) (void)hipStreamDestroy(ctx->xstream);
    if (ctx->ext_ev) (void)hipEventDestroy(ctx->ext_ev);
    if (ctx->ext_block) uc_mbox_retire(ctx->device, ctx->ext_block);
    for (int r = 0; r < P2P_MAX_WORLD; ++r) {
        if (!ctx->p2p_mbox[r]) continue;
        if (ctx->p2p_opened[r]) continue;  // (a peer's mailbox: stays mapped, see ipc_open_cached)
        else if (!host_mbox_release(ctx->p2p_mbox[r])) uc_mbox_retire(ctx->device, ctx->p2p_mbox[r]);
    }
    if (ctx->p2p_err) (void)hipFree(ctx->p2p_err);
    for (auto& pc : ctx->prof)
        for (auto& pr : pc.pending) { (void)hipEventDestroy(pr.a); (void)hipEventDestroy(pr.b); }
    for (auto& pr : ctx->prof_pool) { (void)hipEventDestroy(pr.a); (void)hipEventDestroy(pr.b); }
    if (ctx->board) (void)hipFree(ctx->board);
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->ticket) (void)hipFree(ctx->ticket);
    if (ctx->gran) {
        if (ctx->gran_pooled) {  // (uncached: back to the process-wide pool, see lbfgs_hip_ctx_create)
            std::lock_guard<std::mutex> lk(g_uc_pool_mu);
            g_uc_pool[ctx->device].push_back(ctx->gran);
        } else {
            (void)hipFree(ctx->gran);
        }
    }
    if (ctx->dev_ctr) (void)hipFree(ctx->dev_ctr);
    if (ctx->dot_parts) (void)hipFree(ctx->dot_parts);
    if (ctx->lj_scratch) (void)hipFree(ctx->lj_scratch);
    if (ctx->lj_cells) {
        ctx->lj_cells->release();
        delete ctx->lj_cells;
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->mirror) (void)hipHostFree(ctx->mirror);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int lbfgs_hip_ctx_p2p_seal(lbfgs_hip_ctx* ctx, int* placement_out) {
    if (!ctx) return LBFGS_HIP_ERR_ARG;
    int placement = -1;
    if (ctx->comm_kind == LBFGS_HIP_COMM_P2P) {
        void* own = ctx->p2p_mbox[ctx->shard.rank];
        placement = LBFGS_HIP_MAILBOX_DEVICE;
        std::lock_guard<std::mutex> lk(g_host_mbox_mu);
        auto it = g_host_mbox.find(own);
        if (it != g_host_mbox.end()) {
            placement = LBFGS_HIP_MAILBOX_HOST;
            if (it->second.owner && it->second.linked) {
                (void)shm_unlink(it->second.name.c_str());
                it->second.linked = false;
            }
        }
    }
    if (placement_out) *placement_out = placement;
    return LBFGS_HIP_OK;
}

const char* lbfgs_hip_last_error(const lbfgs_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int lbfgs_hip_sync(lbfgs_hip_ctx* ctx) {
    if (!ctx) return LBFGS_HIP_ERR_ARG;
    return settle(ctx);
}

void* lbfgs_hip_stream(lbfgs_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int lbfgs_hip_get_shard(const lbfgs_hip_ctx* ctx, lbfgs_hip_shard* out) {
    if (!ctx || !out) return LBFGS_HIP_ERR_ARG;
    *out = ctx->shard;
    return LBFGS_HIP_OK;
}

int lbfgs_hip_path_stats(lbfgs_hip_ctx* ctx, uint64_t* resident_two_loops, uint64_t* resident_elements) {
    if (!ctx) return LBFGS_HIP_ERR_ARG;
    if (resident_two_loops) *resident_two_loops = ctx->resident_launches;
    if (resident_elements) *resident_elements = ctx->resident_elements;
    return LBFGS_HIP_OK;
}

int lbfgs_hip_ctx_comm_info(lbfgs_hip_ctx* ctx, lbfgs_hip_comm_info* out) {
    if (!ctx || !out) return LBFGS_HIP_ERR_ARG;
    memset(out, 0, sizeof(*out));
    out->kind = ctx->comm_kind;
    out->world = ctx->shard.world;
    out->rank = ctx->shard.rank;
    out->mailbox_placement = -1;
    out->exclusive_device = ctx->p2p_exclusive ? 1 : 0;
    switch (ctx->comm_kind) {
        case LBFGS_HIP_COMM_RCCL:  // what ncclCommCount / ncclCommUserRank answered when the context was made
            out->ranks_seen = ctx->rccl_ranks_seen;
            out->rank_seen = ctx->rccl_rank_seen;
            break;
        case LBFGS_HIP_COMM_P2P: {  // mailboxes this rank can reach: its own + every peer's it mapped
            out->ranks_seen = 1 + ctx->p2p_peers_device + ctx->p2p_peers_host;
            out->rank_seen = ctx->shard.rank;
            out->peers_device = ctx->p2p_peers_device;
            out->peers_host = ctx->p2p_peers_host;
            std::lock_guard<std::mutex> lk(g_host_mbox_mu);
            out->mailbox_placement = g_host_mbox.count(ctx->p2p_mbox[ctx->shard.rank]) ? LBFGS_HIP_MAILBOX_HOST : LBFGS_HIP_MAILBOX_DEVICE;
            break;
        }
        default:  // none / callback: the library sees no peer itself
            out->ranks_seen = ctx->comm_kind == LBFGS_HIP_COMM_NONE ? 1 : 0;
            out->rank_seen = ctx->shard.rank;
            break;
    }
    out->two_loops = ctx->two_loop_calls;
    out->two_loop_exchanges = ctx->two_loop_exchanges;
    out->allreduce_launches = ctx->allreduce_calls;
    out->p2p_exchanges = ctx->p2p_count;
    out->resident_fallbacks = ctx->resident_fallbacks;
    // the device's own figures (stream.h DevXchg): waits for the stream
    DevXchg x[2];
    const int rc = settle(ctx);
    if (rc != LBFGS_HIP_OK) return rc;
    HIP_TRY(ctx, hipMemcpy(x, reinterpret_cast<const char*>(ctx->dev_ctr) + DEV_XCHG_OFFSET, sizeof(x), hipMemcpyDeviceToHost));
    for (int c = 0; c < 2; ++c) {
        out->timed_exchanges[c] = x[c].count;
        out->exchange_us[c] = (double)x[c].p2p_ticks * 0.01;   // wall_clock64 ticks of 10 ns
        out->local_wait_us[c] = (double)x[c].local_ticks * 0.01;
    }
    static_assert(XCHG_BINS == LBFGS_HIP_XCHG_BINS, "stream.h DevXchgDist and lbfgs_hip_comm_info::exchange_hist");
    DevXchgDist dist[2];
    HIP_TRY(ctx, hipMemcpy(dist, reinterpret_cast<const char*>(ctx->dev_ctr) + DEV_XCHG_DIST_OFFSET, sizeof(dist), hipMemcpyDeviceToHost));
    for (int c = 0; c < 2; ++c) {
        out->exchange_us_max[c] = (double)dist[c].p2p_max * 0.01;
        out->local_wait_us_max[c] = (double)dist[c].local_max * 0.01;
        for (int b = 0; b < XCHG_BINS; ++b) out->exchange_hist[c][b] = dist[c].hist[b];
    }
    return LBFGS_HIP_OK;
}

int lbfgs_hip_set_grid(lbfgs_hip_ctx* ctx, int blocks) {
    if (!ctx || blocks < 0 || blocks > MAX_GRID) return LBFGS_HIP_ERR_ARG;
    ctx->grid_override = blocks;
    return LBFGS_HIP_OK;
}

// ==================================================================================== vectors
int lbfgs_hip_vec_alloc(lbfgs_hip_ctx* ctx, lbfgs_hip_vec** out) {
    if (!ctx || !out) return LBFGS_HIP_ERR_ARG;
    *out = nullptr;
    lbfgs_hip_vec* v = new (std::nothrow) lbfgs_hip_vec();
    if (!v) return fail(ctx, LBFGS_HIP_ERR_NOMEM, "out of host memory");
    v->ctx = ctx;
    size_t bytes = (size_t)ctx->shard.n_local * sizeof(double);
    bytes = (bytes + 255) / 256 * 256;
    if (bytes == 0) bytes = 256;
    // (hipMalloc puts every large vector on a 2 MiB boundary; staggering the bases by 256 B ... 1 MiB per vector changes nothing
    // measurable at n = 1e8 or 1.25e7: profiles/r05_vector_stagger_ab.log)
    hipError_t e = hipMalloc(&v->p, bytes);
    if (e != hipSuccess) {
        delete v;
        return fail(ctx, LBFGS_HIP_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    e = hipMemsetAsync(v->p, 0, bytes, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(v->p);
        delete v;
        return fail(ctx, LBFGS_HIP_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    *out = v;
    return LBFGS_HIP_OK;
}

void lbfgs_hip_vec_free(lbfgs_hip_vec* v) {
    if (!v) return;
    v->ctx->last_res.valid = false;
    (void)hipStreamSynchronize(v->ctx->stream);
    (void)hipFree(v->p);
    delete v;
}

int lbfgs_hip_vec_upload(lbfgs_hip_vec* v, const double* host, uint64_t count) {
    if (!v || (!host && count)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = v->ctx;
    if (count != ctx->shard.n_local) return fail(ctx, LBFGS_HIP_ERR_ARG, "upload of %llu elements into a shard of %llu",
                                                 (unsigned long long)count, (unsigned long long)ctx->shard.n_local);
    if (count == 0) return LBFGS_HIP_OK;
    ctx->last_res.valid = false;  // (an input of the latest resident two-loop may change: it cannot be re-run any more)
    HIP_TRY(ctx, hipMemcpyAsync(v->p, host, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // host buffer may be pageable and reused by the caller
    return LBFGS_HIP_OK;
}

int lbfgs_hip_vec_download(const lbfgs_hip_vec* v, double* host, uint64_t count) {
    if (!v || (!host && count)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = v->ctx;
    if (count != ctx->shard.n_local) return fail(ctx, LBFGS_HIP_ERR_ARG, "download of %llu elements from a shard of %llu",
                                                 (unsigned long long)count, (unsigned long long)ctx->shard.n_local);
    if (count == 0) return LBFGS_HIP_OK;
    // what the stream has produced so far must be sound before it is handed out: a timed-out resident two-loop is re-run with
    // a kernel per step first (its d is an output the caller may be asking for), any other device error is returned
    const int rc_s = settle(ctx);
    if (rc_s != LBFGS_HIP_OK) return rc_s;
    HIP_TRY(ctx, hipMemcpyAsync(host, v->p, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LBFGS_HIP_OK;
}

void* lbfgs_hip_vec_ptr(lbfgs_hip_vec* v) { return v ? (void*)v->p : nullptr; }

int lbfgs_hip_vec_swap(lbfgs_hip_vec* a, lbfgs_hip_vec* b) {
    if (!same_ctx(a, b)) return LBFGS_HIP_ERR_ARG;
    a->ctx->last_res.valid = false;
    std::swap(a->p, b->p);
    return LBFGS_HIP_OK;
}

// ==================================================================================== board
static int device_error(lbfgs_hip_ctx* ctx, unsigned int flag, bool after_rerun = false) {
    ext_abort(ctx);  // (a gated launch that failed: its gates and all-reduces on the second stream are given up with it)
    if (flag == 2u)
        return fail(ctx, LBFGS_HIP_ERR_HIP, "a reduction timed out waiting for a workgroup's partial sums (the latest reducing launch was %s; "
                    "resident two-loops so far %llu, of which re-run per step %llu; resident path %s)",
                    after_rerun ? "the per-step re-run of a resident two-loop" : "not a resident two-loop (or its inputs have changed since)",
                    ctx->resident_launches, ctx->resident_fallbacks, ctx->resident_ok == 1 ? (ctx->resident_proven ? "in use" : "unproven") : "off");
    return fail(ctx, LBFGS_HIP_ERR_COMM, ctx->comm_kind == LBFGS_HIP_COMM_RCCL
                    ? "the gated RCCL exchange of a resident two-loop timed out (no all-reduce answered within the bound)"
                    : "P2P all-reduce timed out waiting for a peer");
}

// one read; *flag receives the device error word that travelled with the results (0 = none)
static int scalars_read_once(lbfgs_hip_ctx* ctx, int first, int count, double* host, unsigned int* flag_out) {
    if (ctx->mirror) {  // fast path: every requested slot was (or is being) published by a kernel's last workgroup
        bool all = true;
        for (int i = 0; i < count && all; ++i) all = ctx->mirror_valid[first + i];
        if (all) {
            const volatile unsigned long long* seq =
                reinterpret_cast<volatile unsigned long long*>(ctx->mirror + LBFGS_HIP_BOARD_SLOTS + 2);
            const unsigned long long want = ctx->mirror_seq;
            bool ok = false;
            for (long spin = 0; spin < 200000000L; ++spin) {  // kernels complete in order: latest seq => all earlier
                if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) >= want) { ok = true; break; }
                if ((spin & 1023) == 1023 && hipStreamQuery(ctx->stream) == hipSuccess) {
                    ok = __atomic_load_n(seq, __ATOMIC_ACQUIRE) >= want;
                    break;
                }
            }
            if (ok) {
                memcpy(host, ctx->mirror + first, count * sizeof(double));
                // a timed-out in-kernel exchange is published next to the results (before the sequence word)
                const volatile unsigned long long* perr =
                    reinterpret_cast<volatile unsigned long long*>(ctx->mirror + LBFGS_HIP_BOARD_SLOTS + 3);
                *flag_out = (unsigned int)*perr;
                return LBFGS_HIP_OK;
            }
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned, ctx->board + first, count * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
    if (ctx->p2p_err)  // pinned[BOARD_SLOTS] is reserved for the P2P timeout flag
        HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned + LBFGS_HIP_BOARD_SLOTS, ctx->p2p_err, sizeof(unsigned int),
                                    hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->pinned, count * sizeof(double));
    if (ctx->p2p_err) {
        unsigned int flag;
        memcpy(&flag, ctx->pinned + LBFGS_HIP_BOARD_SLOTS, sizeof(flag));
        *flag_out = flag;
    }
    return LBFGS_HIP_OK;
}

// The resident two-loop kernel waited in vain for a workgroup: it was not given every CU it asked for (another
// kernel-resident process or stream on this GPU, a CU-masked queue, a partitioned device).  Nothing is lost: its inputs are
// intact (every call that could change them clears last_res).  Clear the error word, never use that kernel again in this
// context, and run the recursion with a kernel per step.
static int recover_resident(lbfgs_hip_ctx* ctx) {
    const lbfgs_hip_ctx::LastResident lr = ctx->last_res;
    ctx->last_res.valid = false;
    ctx->resident_ok = 0;
    ctx->resident_fallbacks += 1;
    fprintf(stderr, "[lbfgs_hip] warning: the on-chip-resident two-loop kernel timed out waiting for a workgroup (the GPU is "
                    "shared with another resident kernel, the device is partitioned, or the queue is CU-masked); re-running this "
                    "two-loop with a kernel per step and staying on that path (LBFGS_HIP_RESIDENT=0 avoids the wait)\n");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->p2p_err, 0, sizeof(unsigned int), ctx->stream));
    // Workgroups of that launch which only STARTED after workgroup 0 had finished (they were never resident together:
    // that is what went wrong) read the counters workgroup 0 had already advanced, and published their partial sums
    // under the tags of launches that are yet to come.  No granule of the aborted launch may survive it.
    HIP_TRY(ctx, hipMemsetAsync(ctx->gran, 0, (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long), ctx->stream));
    if (ctx->mirror) *reinterpret_cast<volatile unsigned long long*>(ctx->mirror + LBFGS_HIP_BOARD_SLOTS + 3) = 0ull;
    int ne = 0;
    return two_loop_eager(lr.h, lr.d, lr.g, lr.k, lr.end, lr.gnum, lr.gden, lr.dn, lr.first, &ne, lr.owl, lr.owl_start, lr.owl_end);
}

}  // extern "C"

// Every entry point that hands results of the stream to the caller without going through the board comes here first
// (lbfgs_hip_sync, lbfgs_hip_vec_download, lbfgs_hip_history_scalars_read): wait for the stream, read the device error
// word, recover from a timed-out resident two-loop (then wait again), return any other error.
int settle(lbfgs_hip_ctx* ctx) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        unsigned int flag = 0;
        if (ctx->p2p_err) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned + LBFGS_HIP_BOARD_SLOTS, ctx->p2p_err, sizeof(unsigned int), hipMemcpyDeviceToHost,
                                        ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            memcpy(&flag, ctx->pinned + LBFGS_HIP_BOARD_SLOTS, sizeof(flag));
        } else {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        if (flag == 0u) {
            if (ctx->last_res.valid) ctx->resident_proven = true;  // (the latest launch was a resident two-loop, and it completed)
            return LBFGS_HIP_OK;
        }
        if (flag == 2u && attempt == 0 && ctx->last_res.valid && ctx->comm_kind == LBFGS_HIP_COMM_NONE) {
            const int rc = recover_resident(ctx);
            if (rc != LBFGS_HIP_OK) return rc;
            continue;
        }
        return device_error(ctx, flag, attempt == 1);
    }
    return LBFGS_HIP_OK;
}

extern "C" {

int lbfgs_hip_scalars_read(lbfgs_hip_ctx* ctx, int first, int count, double* host) {
    if (!ctx || !host || !slot_ok(first, count)) return LBFGS_HIP_ERR_ARG;
    if (count == 0) return lbfgs_hip_sync(ctx);
    unsigned int flag = 0;
    bool rerun = false;
    int rc = scalars_read_once(ctx, first, count, host, &flag);
    if (rc != LBFGS_HIP_OK) return rc;
    if (flag == 2u && ctx->last_res.valid && ctx->comm_kind == LBFGS_HIP_COMM_NONE) {
        if ((rc = recover_resident(ctx)) != LBFGS_HIP_OK) return rc;
        flag = 0;
        rerun = true;
        rc = scalars_read_once(ctx, first, count, host, &flag);
        if (rc != LBFGS_HIP_OK) return rc;
    }
    if (flag) return device_error(ctx, flag, rerun);
    if (ctx->last_res.valid) ctx->resident_proven = true;  // (the read waited for the latest launch: a resident two-loop that completed)
    return LBFGS_HIP_OK;
}

int lbfgs_hip_scalars_write(lbfgs_hip_ctx* ctx, int first, int count, const double* host) {
    if (!ctx || !host || !slot_ok(first, count)) return LBFGS_HIP_ERR_ARG;
    if (count == 0) return LBFGS_HIP_OK;
    for (int i = 0; i < count; ++i) ctx->mirror_valid[first + i] = false;
    ctx->last_res.valid = false;  // (gamma or the first numerator of the latest resident two-loop may change)
    memcpy(ctx->pinned, host, count * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->board + first, ctx->pinned, count * sizeof(double), hipMemcpyHostToDevice,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LBFGS_HIP_OK;
}

void* lbfgs_hip_scalars_ptr(lbfgs_hip_ctx* ctx) { return ctx ? (void*)ctx->board : nullptr; }

int lbfgs_hip_scalars_allreduce(lbfgs_hip_ctx* ctx, int first, int count) {
    if (!ctx || !slot_ok(first, count) || count > MAX_RED) return LBFGS_HIP_ERR_ARG;
    double* ptrs[MAX_RED];
    for (int i = 0; i < count; ++i) ptrs[i] = ctx->board + first + i;
    return allreduce(ctx, ptrs, count);
}

// ==================================================================================== buffers

int lbfgs_hip_host_buffer_create(lbfgs_hip_ctx* ctx, uint64_t bytes, void** out) {
    if (!ctx || !out) return LBFGS_HIP_ERR_ARG;
    *out = nullptr;
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 256, hipHostMallocDefault);
    if (e != hipSuccess)
        return fail(ctx, LBFGS_HIP_ERR_NOMEM, "hipHostMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    *out = p;
    return LBFGS_HIP_OK;
}

void lbfgs_hip_host_buffer_destroy(lbfgs_hip_ctx* ctx, void* buf) {
    if (!buf) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    (void)hipHostFree(buf);
}

int lbfgs_hip_device_buffer_create(lbfgs_hip_ctx* ctx, const void* host, uint64_t bytes, void** out) {
    if (!ctx || !out || (!host && bytes)) return LBFGS_HIP_ERR_ARG;
    *out = nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 256);
    if (e != hipSuccess) return fail(ctx, LBFGS_HIP_ERR_NOMEM, "hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    if (bytes) {
        e = hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(ctx, LBFGS_HIP_ERR_HIP, "upload: %s", hipGetErrorString(e));
        }
    }
    *out = p;
    return LBFGS_HIP_OK;
}

void lbfgs_hip_device_buffer_destroy(lbfgs_hip_ctx* ctx, void* buf) {
    if (!buf) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(buf);
}

// ==================================================================================== measurement
int lbfgs_hip_prof_enable(lbfgs_hip_ctx* ctx, int on) {
    if (!ctx) return LBFGS_HIP_ERR_ARG;
    ctx->prof_on = on != 0;
    return LBFGS_HIP_OK;
}


int lbfgs_hip_prof_reset(lbfgs_hip_ctx* ctx) {
    if (!ctx) return LBFGS_HIP_ERR_ARG;
    int rc = prof_drain(ctx);
    for (auto& pc : ctx->prof) {
        pc.launches = 0;
        pc.ms = 0.0;
    }
    return rc;
}

int lbfgs_hip_prof_read(lbfgs_hip_ctx* ctx, int kclass, uint64_t* launches, double* total_ms) {
    if (!ctx || kclass < 0 || kclass >= LBFGS_HIP_K_CLASSES) return LBFGS_HIP_ERR_ARG;
    int rc = prof_drain(ctx);
    if (launches) *launches = ctx->prof[kclass].launches;
    if (total_ms) *total_ms = ctx->prof[kclass].ms;
    return rc;
}

}  // extern "C"

