// rust-lbfgs_amd/csrc/lbfgs_hip.hip -- implementation of include/lbfgs_hip.h for gfx950 (MI355X).
//
// Context (device, stream, scalar board, reduction scratch, communicator), device vectors,
// and one launch wrapper per C-ABI entry point.  The kernels are instantiations of the
// streaming skeleton in stream.h over the element operators in ops.h.
//
// There is deliberately NO CPU fallback: without a GPU lbfgs_hip_ctx_create() fails with
// LBFGS_HIP_ERR_NO_DEVICE.
#include "../../include/lbfgs_hip.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <map>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ops.h"
#include "gram.h"
#include "lj.h"
#include "resident.h"
#include "gram_combine.h"

using namespace lh;

// ------------------------------------------------------------------------------------ RCCL (lazy)
// RCCL is only needed when world > 1, so it is dlopen'ed on first use; a single-GPU process
// never loads it.
namespace {
typedef struct { char internal[128]; } nccl_unique_id_t;
typedef void* nccl_comm_t;
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id_t*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id_t, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*CommCount)(const nccl_comm_t, int*) = nullptr;     // what RCCL itself says the communicator spans ...
    int (*CommUserRank)(const nccl_comm_t, int*) = nullptr;  // ... and who this process is in it
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
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
        x.p2p_ticks += (unsigned long long)(wall_clock64() - t1);
        x.count += 1ull;
        *xs = x;
    }
}

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
    unsigned long long resident_attr_mask = 0;  // which resident kernels have had their dynamic-LDS limit raised on this device
    int gram_combine_resident = 1;        // LBFGS_HIP_GRAM_COMBINE_RESIDENT=0: the vector-free combine as one streaming pass over all columns
    int lj_build_fp32 = 1;                // LBFGS_HIP_LJ_BUILD_FP32=0: the LJ_CELLS list from double-precision candidate tests
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
    int resident_fault = 0;               // LBFGS_HIP_RESIDENT_FAULT=k (tests): the k-th resident launch of this context loses its last workgroup
    bool defer_inner_sums = true;         // LBFGS_HIP_DEFER_SUMS=0: the two-loop's inner dots are reduced by their own kernels (A/B)
    double* dot_parts = nullptr;          // 2 x MAX_GRID: workgroup partials of the two-loop's inner dot products (ping-pong)
    DevCounters* dev_ctr = nullptr;       // device-resident sequence numbers (stream.h); the three fields below shadow them
    unsigned long long red_count = 0;     // tagged reducing launches enqueued so far: the next one uses tag (red_count % (2^32-1)) + 1
    // hipGraph replay of the two-loop recursion (lbfgs_hip_two_loop*).  OFF by default: measured on MI355X / ROCm 7.2
    // (profiles/r02_graph_vs_eager.log) one hipGraphLaunch of the 2m-kernel chain is 0-2 % SLOWER than 2m eager launches
    // at every size from n = 1e5 to 1.25e7 -- the chain is bound by the GPU-side cost of a dependent kernel boundary,
    // not by the host's launches.  LBFGS_HIP_GRAPH=1 (or LBFGS_HIP_GRAPH_MAX_MB) turns it on.
    size_t graph_max_bytes = 0;           // vectors up to this size replay their two-loop as a graph (0 = never)
    bool capturing = false;
    std::vector<std::pair<int, bool>> capture_touch;  // mirror_valid assignments made while capturing
    int grid_class[16] = {0};             // LBFGS_HIP_GRID_K<class>=N: grid override per kernel class (in-situ tuning)
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
    int gram_grid = 0;  // workgroups of the Gram rows kernel (0 = same as the others)
    bool gram_grid_forced = false;
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

// One recorded two-loop recursion: the hipGraph of its kernels plus what launching them does to the host-side shadows.
struct TwoLoopGraph {
    hipGraphExec_t exec = nullptr;
    unsigned long long tagged_launches = 0, mirrored_launches = 0, p2p_exchanges = 0;
    std::vector<std::pair<int, bool>> mirror_touch;
    int new_end = 0;
};
typedef std::array<uint64_t, 16> TwoLoopKey;

struct lbfgs_hip_history {
    lbfgs_hip_ctx* ctx;
    int m;
    std::map<TwoLoopKey, TwoLoopGraph> graphs;  // keyed by everything that shapes the launch sequence (two_loop_impl)
    std::vector<lbfgs_hip_vec*> s, y;
    double* ys;     // device, m
    double* alpha;  // device, m
    // vector-free (Gram) two-loop: G = B^T B over the basis [s.., y.., g], refreshed rows, coefficients
    double* gram = nullptr;        // (2m+1)^2
    double* gram_rows = nullptr;   // 3*(2m+1), contiguous (one all-reduce message)
    double* gram_delta = nullptr;  // 2m+1
    double* gram_pred = nullptr;   // 1: ||d||^2 predicted by the coefficient-space arithmetic (gram_coef_kernel)
};

namespace {

int fail(lbfgs_hip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(ctx, LBFGS_HIP_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

inline bool slot_ok(int first, int count) {
    return first >= 0 && count >= 0 && first + count <= LBFGS_HIP_BOARD_SLOTS;
}

int grid_for(const lbfgs_hip_ctx* ctx, int x32 = 27) {
    int g = ctx->grid_override > 0 ? ctx->grid_override : (x32 == 27 ? ctx->grid_default : std::max(1, ctx->cu_count * x32 / 32));
    if (g > MAX_GRID) g = MAX_GRID;
    if (g < 1) g = 1;
    return g;
}

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
            if (ctx->capturing) return fail(ctx, LBFGS_HIP_ERR_HIP, "tag wrap inside a graph capture");
            HIP_TRY(ctx, hipMemsetAsync(ctx->gran, 0, (size_t)MAX_RED * MAX_GRID * 2 * sizeof(unsigned long long), ctx->stream));
        }
        ctx->red_count += 1;
    }
    return LBFGS_HIP_OK;
}

// the P2P control block of the NEXT reduction (epochs advance identically on every rank)
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

#ifndef LH_NT_IN   // cache-hint masks of the streaming instantiation (bit s = stream s uses `nt`); A/B builds override them
#define LH_NT_IN ~0u
#endif
#ifndef LH_NT_OUT
#define LH_NT_OUT ~0u
#endif
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
                if (ctx->capturing) ctx->capture_touch.emplace_back((int)idx, mirrored);
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

// ---- launch one operator ---------------------------------------------------------------------
// partials_only (Op::NRED == 1, single rank): the kernel leaves its workgroups' partial sums in red_out[0][0 .. grid) and
// nobody reduces them -- the consumer does (stream.h RedCtl::tagged == 2); *grid_out receives the number of partials.
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
    const int grid = (kclass >= 0 && kclass < 16 && ctx->grid_class[kclass] > 0)
                         ? std::min(MAX_GRID, ctx->grid_class[kclass]) : grid_for(ctx, tuning<Op>::GRID_X32);
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

inline bool same_ctx(const lbfgs_hip_vec* a, const lbfgs_hip_vec* b) { return a && b && a->ctx == b->ctx; }

}  // namespace

// ---- vector-free (Gram) two-loop: scalar recursion on the coefficients --------------------------------
constexpr int GRAM_MAX_M = 10;  // 3*(2m+1) sums must fit MAX_RED
constexpr int GRAM_MAX_NB = 2 * GRAM_MAX_M + 1;
struct GramArgs {
    double* G;             // nb x nb, basis order [s_0..s_{m-1}, y_0..y_{m-1}, g]
    const double* rows;    // 3 x nb in POSITION order of the rows kernel
    double* delta;         // nb coefficients out (basis order)
    double* pred;          // out: ||d||^2 = delta^T G delta as the GRAM arithmetic sees it (NaN if a coefficient is not finite):
                           // the combine pass sums the real ||d||^2; a mismatch says the coefficient-space recursion has lost its
                           // digits (lbfgs_hip_two_loop_gram, board[dnorm_slot + 2])
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
        *a.pred = finite ? dd : __longlong_as_double(0x7ff8000000000000LL);
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

int lj_cells_eval(lbfgs_hip_ctx* ctx, const lbfgs_hip_objective* obj, const lbfgs_hip_vec* x, lbfgs_hip_vec* g, double* out,
                  uint32_t natoms) {
    int rc = lj_cells_prepare(ctx, obj, natoms);
    if (rc != LBFGS_HIP_OK) return rc;
    LjCells* lc = ctx->lj_cells;
    if (!lc->built && (rc = lj_cells_rebuild(ctx, x->p)) != LBFGS_HIP_OK) return rc;
    const double rc2 = obj->cutoff * obj->cutoff;
    const double i6 = 1.0 / (rc2 * rc2 * rc2);
    const double eshift = 4.0 * (i6 * i6 - i6);
    const double half_skin2 = 0.25 * obj->skin * obj->skin;
    const uint32_t want = (natoms + BLOCK - 1) / BLOCK;
    const uint32_t grid = std::max(1u, std::min(want, (uint32_t)MAX_GRID));
    for (int attempt = 0;; ++attempt) {
        RedCtl red{};
        if ((rc = fill_handoff(ctx, red, 2)) != LBFGS_HIP_OK) return rc;
        red.out[0] = out;
        red.out[1] = lc->host_dev;  // the "moved too far" count goes straight to host-mapped memory
        // a cheap look first: has the list gone stale?  (then the evaluation kernel returns at once and the list is rebuilt)
        unsigned int* moved_flag = lc->overflow + 1;  // (a word of the 64-byte scratch next to the overflow counter)
        HIP_TRY(ctx, hipMemsetAsync(moved_flag, 0, sizeof(unsigned int), ctx->stream));
        hipLaunchKernelGGL(lj_cells_check_kernel, dim3(std::min(grid, 1024u)), dim3(BLOCK), 0, ctx->stream, x->p, lc->xref, natoms,
                           half_skin2, moved_flag);
        // Which table?  The list kernel writes it row-major; turning it costs 0.17 ms at 1e6 atoms and makes every evaluation
        // 0.05 ms cheaper.  A list that is rebuilt after two or three evaluations (a system that relaxes fast) is read as
        // written; one that has served LJ_TRANSPOSE_AFTER evaluations -- or whose predecessor did -- is turned.
        if (!lc->transposed && lc->evals_this_list >= LJ_TRANSPOSE_AFTER) lj_cells_transpose(ctx, lc);
        if (lc->transposed)
            hipLaunchKernelGGL(lj_cells_eval_kernel<false>, dim3(grid), dim3(BLOCK), 0, ctx->stream, x->p, g->p, lc->nbr, lc->cnt,
                               lc->xref, natoms, lc->max_nbr, rc2, eshift, half_skin2, moved_flag, red);
        else
            hipLaunchKernelGGL(lj_cells_eval_kernel<true>, dim3(grid), dim3(BLOCK), 0, ctx->stream, x->p, g->p, lc->nbr_rows, lc->cnt,
                               lc->xref, natoms, lc->max_nbr, rc2, eshift, half_skin2, moved_flag, red);
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

// Lennard-Jones objectives (lj.h): x holds 3*natoms coordinates of ONE rank
int lj_eval(lbfgs_hip_ctx* ctx, const lbfgs_hip_objective* obj, const lbfgs_hip_vec* x, lbfgs_hip_vec* g, double* out) {
    if (ctx->shard.world != 1) return fail(ctx, LBFGS_HIP_ERR_ARG, "the LJ objectives need all atoms on one rank");
    const uint64_t n = ctx->shard.n_local;
    if (n % 3 != 0 || n / 3 > 0x7fffffffULL) return fail(ctx, LBFGS_HIP_ERR_ARG, "LJ needs n = 3*natoms");
    const uint32_t natoms = (uint32_t)(n / 3);
    const long idx = out - ctx->board;
    if (idx >= 0 && idx < LBFGS_HIP_BOARD_SLOTS + 2) ctx->mirror_valid[idx] = false;
    if (obj->kind == LBFGS_HIP_OBJ_LJ_CELLS) {
        ProfScope ps(ctx, LBFGS_HIP_K_EVAL);
        return lj_cells_eval(ctx, obj, x, g, out, natoms);
    }
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

int rosen_shape_ok(lbfgs_hip_ctx* ctx) {
    if ((ctx->shard.n_local & 1) || (ctx->shard.offset & 1))
        return fail(ctx, LBFGS_HIP_ERR_ARG, "Rosenbrock couples (x[2i], x[2i+1]): shard size and offset must be even");
    return LBFGS_HIP_OK;
}
}  // namespace

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

namespace {
// the rows pass with the row vectors' tiles on the chip (gram_combine.h): -> true if launched (red is set up by the caller)
bool gram_rows_resident(lbfgs_hip_ctx* ctx, const double* const* in, int nb, const RedCtl& red) {
    if (!ctx->resident_on || !ctx->gram_combine_resident || ctx->capturing || ctx->grid_override > 0 || ctx->gram_grid_forced ||
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
    if (!ctx->resident_on || !ctx->gram_combine_resident || ctx->handoff_ticket || ctx->capturing || ctx->grid_override > 0 ||
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
    a.lead = ctx->shard.offset == 0 ? 1 : 0;  // (the rank that owns element 0 carries the prediction into the third sum)
    a.total_rounds = (uint32_t)total;
    a.tile_rounds = (uint32_t)tile;
    RedCtl red{};
    bool in_kernel_exchange = false;
    double* outs2[3] = {dn, dn + 1, dn + 2};
    const int rc_p = prep_red(ctx, red, 3, outs2, nullptr, 0, &in_kernel_exchange);
    if (rc_p != LBFGS_HIP_OK) return rc_p;
    {
        ProfScope ps(ctx, LBFGS_HIP_K_TWOLOOP_STEP);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds_bytes, ctx->stream, a, red);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (!in_kernel_exchange) {
        const int rc = allreduce(ctx, outs2, 3);
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
    double* outs2[3] = {dn, dn + 1, dn + 2};
    return launch(ctx, LBFGS_HIP_K_TWOLOOP_STEP, cmb, outs2);
}
}  // namespace

static int two_loop_impl(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                         int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                         bool owl, uint64_t owl_start, uint64_t owl_end);
static int two_loop_eager(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                          int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                          bool owl, uint64_t owl_start, uint64_t owl_end);

#if LH_RES_TRACE
namespace { void res_trace_print(); }
#endif
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
    if (const char* e = getenv("LBFGS_HIP_HANDOFF_TIMEOUT_MS")) ctx->handoff_timeout_ticks = (unsigned long long)std::max(1, atoi(e)) * 100000ULL;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_FIRST_TIMEOUT_MS")) ctx->first_timeout_ticks = (unsigned long long)std::max(1, atoi(e)) * 100000ULL;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_NT_MB")) ctx->resident_nt_bytes = (size_t)std::max(0, atoi(e)) << 20;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_HYBRID")) ctx->resident_hybrid = atoi(e) != 0;
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_TOUCH")) ctx->resident_touch = std::min(LH_RES_TOUCH, std::max(-1, atoi(e)));
    if (const char* e = getenv("LBFGS_HIP_RESIDENT_PLAIN_MB")) ctx->resident_plain_bytes = (size_t)std::max(0, atoi(e)) << 20;
    if (const char* e = getenv("LBFGS_HIP_GRAPH")) ctx->graph_max_bytes = atoi(e) ? ~(size_t)0 : 0;
    if (const char* e = getenv("LBFGS_HIP_GRAPH_MAX_MB")) ctx->graph_max_bytes = (size_t)atoll(e) << 20;
    if (const char* e = getenv("LBFGS_HIP_NT_STORE_THRESHOLD_MB")) ctx->nt_store_threshold_bytes = (size_t)atoll(e) << 20;
    if (const char* e = getenv("LBFGS_HIP_GRAM_GRID")) {  // (asking for a grid asks for the streaming rows kernel)
        ctx->gram_grid = std::min(MAX_GRID, std::max(0, atoi(e)));
        ctx->gram_grid_forced = true;
    }
    for (int k = 0; k < LBFGS_HIP_K_CLASSES && k < 16; ++k) {
        char name[32];
        snprintf(name, sizeof(name), "LBFGS_HIP_GRID_K%d", k);
        if (const char* e = getenv(name)) ctx->grid_class[k] = std::max(0, atoi(e));
    }
    if (const char* e = getenv("LBFGS_HIP_HANDOFF")) ctx->handoff_ticket = strcmp(e, "ticket") == 0;
    if (const char* e = getenv("LBFGS_HIP_GRID")) ctx->grid_override = std::min(MAX_GRID, std::max(0, atoi(e)));
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
            hipError_t e = (fail_hook && atoi(fail_hook) != 0) ? hipErrorInvalidValue : ipc_open_cached(hdl, &p);
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
    if (ctx->nccl && g_rccl.ok) g_rccl.CommDestroy(ctx->nccl);
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

static int settle(lbfgs_hip_ctx* ctx);  // (below, with the board: waits for the stream and looks at the device error word)

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
    double* t = a->p;
    a->p = b->p;
    b->p = t;
    return LBFGS_HIP_OK;
}

// ==================================================================================== board
static int device_error(lbfgs_hip_ctx* ctx, unsigned int flag) {
    if (flag == 2u)
        return fail(ctx, LBFGS_HIP_ERR_HIP, "a reduction timed out waiting for a workgroup's partial sums");
    return fail(ctx, LBFGS_HIP_ERR_COMM, "P2P all-reduce timed out waiting for a peer");
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

// Every entry point that hands results of the stream to the caller without going through the board comes here first
// (lbfgs_hip_sync, lbfgs_hip_vec_download, lbfgs_hip_history_scalars_read): wait for the stream, read the device error
// word, recover from a timed-out resident two-loop (then wait again), return any other error.
static int settle(lbfgs_hip_ctx* ctx) {
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
        return device_error(ctx, flag);
    }
    return LBFGS_HIP_OK;
}

int lbfgs_hip_scalars_read(lbfgs_hip_ctx* ctx, int first, int count, double* host) {
    if (!ctx || !host || !slot_ok(first, count)) return LBFGS_HIP_ERR_ARG;
    if (count == 0) return lbfgs_hip_sync(ctx);
    unsigned int flag = 0;
    int rc = scalars_read_once(ctx, first, count, host, &flag);
    if (rc != LBFGS_HIP_OK) return rc;
    if (flag == 2u && ctx->last_res.valid && ctx->comm_kind == LBFGS_HIP_COMM_NONE) {
        if ((rc = recover_resident(ctx)) != LBFGS_HIP_OK) return rc;
        flag = 0;
        rc = scalars_read_once(ctx, first, count, host, &flag);
        if (rc != LBFGS_HIP_OK) return rc;
    }
    if (flag) return device_error(ctx, flag);
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
    for (auto& kv : h->graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
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

// -> 1 if the recursion was launched as the resident kernel, 0 if this case is not eligible (caller falls back), < 0 error
int two_loop_resident(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                      int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end, bool owl,
                      uint64_t owl_start, uint64_t owl_end) {
    lbfgs_hip_ctx* ctx = h->ctx;
    const int m = h->m;
    const int bound = (int)std::min<uint64_t>((uint64_t)m, k);
    // One rank -- or several that each have their GPU to themselves: a kernel that fills the chip and waits for its peers
    // inside would starve ranks sharing the GPU (lbfgs_hip_comm.exclusive_device).
    const bool comm_ok = ctx->comm_kind == LBFGS_HIP_COMM_NONE || (ctx->comm_kind == LBFGS_HIP_COMM_P2P && ctx->p2p_exclusive);
    if (!ctx->resident_on || !comm_ok || ctx->handoff_ticket || bound < 1 || ctx->capturing || ctx->grid_override > 0 ||
        2 * bound > RES_MAX_STEPS)
        return 0;
    // one workgroup per CU: all of them resident at once (fewer on request -- tests run two ranks on one GPU -- and for
    // vectors of a few MB, where a hand-off among fewer workgroups is worth more than the idle CUs' bandwidth: >= 8 pairs per
    // thread, at least 64 workgroups; measured at n = 1e5 / 3e5 / 1e6, profiles/r02_resident_small_n.log)
    const uint64_t n = ctx->shard.n_local;
    // ... and 27/32 of the CUs, the streaming kernels' grid, once less than a fifth of a shard fits the chip and
    // the kernel is a streaming kernel above all (n = 1e8: 9.73 ms against 9.83 with 256; at 2.5e7 256 wins by 4 %)
    const bool mostly_streaming = (n >> 1) > 5ull * (uint64_t)(60 + RES_LDS_PAIRS_MAX) * (uint64_t)ctx->cu_count * BLOCK;
    const int grid_auto = mostly_streaming ? std::max(1, ctx->cu_count * 27 / 32)
        : (int)std::min<uint64_t>((uint64_t)ctx->cu_count, std::max<uint64_t>(64, ((n >> 1) + BLOCK * 8 - 1) / (BLOCK * 8)));
    const int grid = ctx->resident_grid > 0 ? std::min(ctx->resident_grid, ctx->cu_count) : grid_auto;
    if (grid < 1 || grid > BLOCK || grid > MAX_GRID) return 0;  // (every thread polls one workgroup's granules)
    const uint64_t per_round = (uint64_t)grid * BLOCK;
    const uint64_t E = ((n >> 1) + per_round - 1) / per_round;  // 16-byte pairs per thread
    constexpr int ER_MAX = 60;
    // Larger shards: HYBRID -- the first 60 + 36 rounds of every thread stay on the chip, the rest of q lives in `d` and is
    // streamed by every step as on the kernel-per-step path (resident.h).
    const bool hybrid = E > (uint64_t)(ER_MAX + RES_LDS_PAIRS_MAX);
    if (hybrid && !ctx->resident_hybrid) return 0;
    if (E == 0 || (n >> 1) + per_round * 4 >= (1ull << 28)) return 0;
    if (ctx->red_count % 0xFFFFFFFFull + 2ull * (uint64_t)bound + 4ull >= 0xFFFFFFFFull) return 0;  // tag wrap: eager path
    // Rounds 0 .. E-2 are full for every thread, round E-1 is the ragged one.  The register rounds carry no bounds checks,
    // so ER <= E-1; the rest (the ragged round included) lives in LDS -- everything, for the smallest vectors (ER = 0).
    const int er = E - 1 >= ER_MAX ? ER_MAX : E - 1 >= 40 ? 40 : E - 1 >= 24 ? 24 : E - 1 >= 8 ? 8 : 0;
    const uint32_t eh = hybrid ? (uint32_t)(E - ER_MAX - RES_LDS_PAIRS_MAX) : 0u;  // rounds whose q stays in HBM
    const uint32_t el = hybrid ? (uint32_t)RES_LDS_PAIRS_MAX : (uint32_t)((E - er + RES_UNROLL - 1) / RES_UNROLL * RES_UNROLL);
    if (el > (uint32_t)RES_LDS_PAIRS_MAX) return 0;
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
    if (ctx->resident_ok != 1) return 0;

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
    ra.touch_rounds = (uint32_t)(ctx->resident_touch >= 0 ? ctx->resident_touch : (E >= 64 ? 16 : 8));
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
    // until a resident launch of this context has been seen to complete, a missing workgroup costs milliseconds (see
    // lbfgs_hip_ctx::first_timeout_ticks); one rank only: with peers a hand-off also waits for THEIR start-up
    if (!ctx->resident_proven && ctx->comm_kind == LBFGS_HIP_COMM_NONE)
        red.timeout_ticks = std::min(red.timeout_ticks, ctx->first_timeout_ticks);
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
#if LH_RES_TRACE
    res_trace_collect(ctx, trace_first_tag, (int)handoffs, grid);
#endif
    ctx->resident_launches += 1;
    ctx->resident_elements = hybrid ? 2ull * per_round * (uint64_t)(er + el) : n;
    *new_end = e1;
    return 1;
}
}  // namespace

static int two_loop_eager(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
                          int gamma_num_slot, int gamma_den_slot, int dnorm_slot, int first_dot_slot, int* new_end,
                          bool owl, uint64_t owl_start, uint64_t owl_end);

// The recursion is a chain of 2*bound kernels whose arguments depend only on (ring position, bound, operand addresses,
// slots): with the launch-to-launch sequence numbers in device memory (stream.h DevCounters) the chain can be recorded
// once per such combination and replayed with ONE hipGraphLaunch -- the host stops paying a launch per kernel, which
// is what bounds iterations/s on vectors of a few MB.  Recording = stream capture of the eager path, so both paths
// launch the very same kernels with the very same arguments (results are bitwise equal; a test checks it).
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
    const uint64_t bound = std::min<uint64_t>((uint64_t)h->m, k);
    const bool graphable = bound > 0 && !ctx->prof_on && !ctx->capturing &&
                           (size_t)ctx->shard.n_local * sizeof(double) <= ctx->graph_max_bytes &&
                           (ctx->comm_kind == LBFGS_HIP_COMM_NONE || ctx->comm_kind == LBFGS_HIP_COMM_P2P) &&
                           // (a tag wrap needs a memset between two launches: take the eager path across it)
                           ctx->red_count % 0xFFFFFFFFull + 4ull * (uint64_t)h->m + 8ull < 0xFFFFFFFFull;
    if (!graphable)
        return two_loop_eager(h, d, g, k, end, gamma_num_slot, gamma_den_slot, dnorm_slot, first_dot_slot, new_end, owl,
                              owl_start, owl_end);
    uint64_t hp = 1469598103934665603ull;  // FNV-1a over the history's vector addresses (they never change; cheap insurance)
    for (int j = 0; j < h->m; ++j)
        for (const double* p : {h->s[j]->p, h->y[j]->p}) hp = (hp ^ (uint64_t)(uintptr_t)p) * 1099511628211ull;
    const TwoLoopKey key = {(uint64_t)end, bound, (uint64_t)(uintptr_t)g->p, (uint64_t)(uintptr_t)d->p, (uint64_t)gamma_num_slot,
                            (uint64_t)gamma_den_slot, (uint64_t)dnorm_slot, (uint64_t)(int64_t)first_dot_slot, (uint64_t)owl,
                            owl_start, owl_end, (uint64_t)ctx->grid_override, hp, (uint64_t)ctx->handoff_ticket | ((uint64_t)ctx->defer_inner_sums << 1),
                            (uint64_t)ctx->shard.n_local, (uint64_t)ctx->nt_threshold_bytes ^ ((uint64_t)ctx->nt_store_threshold_bytes << 1)};
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        // record: the eager path under stream capture (nothing executes; the host-side shadows advance as usual)
        TwoLoopGraph tg;
        const unsigned long long r0 = ctx->red_count, m0 = ctx->mirror_seq, p0 = ctx->p2p_count;
        ctx->capture_touch.clear();
        HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        ctx->capturing = true;
        int ne = end;
        const int rc = two_loop_eager(h, d, g, k, end, gamma_num_slot, gamma_den_slot, dnorm_slot, first_dot_slot, &ne, owl,
                                      owl_start, owl_end);
        ctx->capturing = false;
        hipGraph_t graph = nullptr;
        const hipError_t e_end = hipStreamEndCapture(ctx->stream, &graph);
        if (rc != LBFGS_HIP_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e_end != hipSuccess || !graph) return fail(ctx, LBFGS_HIP_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e_end));
        const hipError_t e_inst = hipGraphInstantiate(&tg.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e_inst != hipSuccess) return fail(ctx, LBFGS_HIP_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e_inst));
        tg.tagged_launches = ctx->red_count - r0;
        tg.mirrored_launches = ctx->mirror_seq - m0;
        tg.p2p_exchanges = ctx->p2p_count - p0;
        tg.mirror_touch = ctx->capture_touch;
        tg.new_end = ne;
        it = h->graphs.emplace(key, std::move(tg)).first;
    } else {
        const TwoLoopGraph& tg = it->second;  // replay: what the launches do to the shadows, then the launches themselves
        ctx->red_count += tg.tagged_launches;
        ctx->mirror_seq += tg.mirrored_launches;
        ctx->p2p_count += tg.p2p_exchanges;
        ctx->two_loop_exchanges += tg.p2p_exchanges;
        for (const auto& t : tg.mirror_touch) ctx->mirror_valid[t.first] = t.second;
    }
    *new_end = it->second.new_end;
    HIP_TRY(ctx, hipGraphLaunch(it->second.exec, ctx->stream));
    return LBFGS_HIP_OK;
}

static int two_loop_eager(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end,
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
    if (!slot_ok(gamma_num_slot, 1) || !slot_ok(gamma_den_slot, 1) || !slot_ok(dnorm_slot, 3)) return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = h->ctx;
    const int m = h->m;
    if (m > GRAM_MAX_M) return fail(ctx, LBFGS_HIP_ERR_ARG, "vector-free two-loop supports m <= %d", GRAM_MAX_M);
    if (!h->gram) {
        const size_t nb = 2 * (size_t)m + 1, words = nb * nb + 3 * nb + nb + 1;
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
        case LBFGS_HIP_OBJ_LJ_ALLPAIRS:
        case LBFGS_HIP_OBJ_LJ_NEIGHBORS:
        case LBFGS_HIP_OBJ_LJ_CELLS: {  // gather pattern: line step, evaluate, g.d as three launches
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

int lbfgs_hip_objective_owlqn_line_eval(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                        const lbfgs_hip_vec* d, double step, const lbfgs_hip_vec* wp, lbfgs_hip_vec* g,
                                        lbfgs_hip_vec* pg, double c, uint64_t start, uint64_t end, int out_slot) {
    if (!obj || !same_ctx(x, xp) || !same_ctx(x, d) || !same_ctx(x, wp) || !same_ctx(x, g) || !same_ctx(x, pg) ||
        !slot_ok(out_slot, 5))
        return LBFGS_HIP_ERR_ARG;
    lbfgs_hip_ctx* ctx = x->ctx;
    double* b = ctx->board + out_slot;
    double* outs[5] = {b, b + 1, b + 2, b + 3, b + 4};
    switch (obj->kind) {
        case LBFGS_HIP_OBJ_QUADRATIC: {
            OpObjOwlLineEval<ObjQuadratic> op{};
            op.in[0] = xp->p; op.in[1] = d->p; op.in[2] = wp->p; op.out[0] = x->p; op.out[1] = g->p; op.out[2] = pg->p;
            op.step = step; op.c = c; op.start = start; op.end = end; op.obj = {obj->seed_a, obj->seed_b};
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        case LBFGS_HIP_OBJ_LOGISTIC: {
            OpObjOwlLineEval<ObjLogistic> op{};
            op.in[0] = xp->p; op.in[1] = d->p; op.in[2] = wp->p; op.out[0] = x->p; op.out[1] = g->p; op.out[2] = pg->p;
            op.step = step; op.c = c; op.start = start; op.end = end; op.obj = {obj->seed_a, obj->seed_b};
            return launch(ctx, LBFGS_HIP_K_EVAL, op, outs);
        }
        default: {  // objectives without a fused form: the three separate passes
            int rc = lbfgs_hip_line_step(x, xp, d, step, wp, start, end);
            if (rc != LBFGS_HIP_OK) return rc;
            if ((rc = lbfgs_hip_objective_eval(obj, x, g, out_slot)) != LBFGS_HIP_OK) return rc;
            if ((rc = lbfgs_hip_owlqn_post_eval(x, g, pg, c, start, end, out_slot + 2)) != LBFGS_HIP_OK) return rc;
            return lbfgs_hip_vecdot(g, d, out_slot + 1);
        }
    }
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
