// tests/support/mock_lbfgs_hip.cpp -- TEST DOUBLE of include/lbfgs_hip.h on the CPU.
//
// TEST INFRASTRUCTURE ONLY.  It exists so that the host orchestration
// (rust-lbfgs_amd/csrc/host/solver.cpp) -- line-search state machines, stop tests,
// error paths, sharding and the all-reduce plumbing -- can be exercised in the
// `-m "not gpu"` suite, where there is no GPU.  It is linked ONLY into
// tests/support/liblbfgs_solver_mock.so, never into the product libraries, and the
// product never falls back to it: rust-lbfgs_amd/liblbfgs_hip.so has no CPU path.
//
// Every operation is composed from the CPU oracle's primitives (oracle/) in the
// reference's unfused order with sequential sums, so with world == 1 the solver on
// this mock must reproduce the oracle -- and the reference's 17-digit known answers
// -- bit for bit.  With world > 1 each rank holds a contiguous shard and scalars are
// closed through the callback communicator (torch.distributed / gloo in the tests) -- or, round 6, through STAND-INS of the
// P2P and RCCL communicators over POSIX shared memory ("bus" below): several rank PROCESSES, which no single GPU can host,
// then run bench.py's legs, dist.py's collective fall-back from device- to host-placed mailboxes, and -- through the product's own
// collective skeleton, rust-lbfgs_amd/csrc/ext_protocol.h -- the preparation of the gated RCCL exchange with failures injected
// on ONE rank.  Hooks (environment): LBFGS_MOCK_RCCL=1 (there is an "RCCL"), LBFGS_MOCK_NO_DEVICE_IPC=1 (a peer's device-placed
// mailbox cannot be mapped: each rank sees only its own GPU), LBFGS_MOCK_FAULT_RANK=r (LBFGS_HIP_RESIDENT_FAULT applies to
// rank r only), LBFGS_MOCK_HANDSHAKE_FAIL_RANK=r, LBFGS_MOCK_PROTOCOL_LOG=<path prefix> (every rank logs its collectives).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lbfgs_hip.h"
#include "../../oracle/lbfgs_oracle.h"
#include "../../rust-lbfgs_amd/csrc/ext_protocol.h"

// ---- the bus: a ring of slots [BUS_RING][world], one per (epoch mod ring, writing rank), in shared memory -------------------
constexpr int BUS_RING = 8, BUS_COUNT = 8, BUS_WORLD = 8;
struct BusSlot {
    std::atomic<uint64_t> epoch;  // written last (release); 0 = never
    uint64_t tag;                 // what the writer thinks this collective is (count and kind): a mismatch is a protocol error
    double v[BUS_COUNT];
};
struct BusSeg {
    BusSlot* slots = nullptr;  // [BUS_RING][BUS_WORLD]
    std::string name;
    bool owner = false;
    int placement = -1;        // P2P mailboxes: LBFGS_HIP_MAILBOX_DEVICE / _HOST as ASKED for (it is shared memory either way)
};
constexpr size_t BUS_BYTES = sizeof(BusSlot) * BUS_RING * BUS_WORLD;
constexpr char BUS_MAGIC[8] = {'L', 'M', 'O', 'C', 'K', 'B', 'U', 'S'};

struct lbfgs_hip_ctx {
    lbfgs_hip_shard shard{};
    int comm_kind = LBFGS_HIP_COMM_NONE;
    lbfgs_hip_allreduce_cb cb = nullptr;
    void* cb_user = nullptr;
    double board[LBFGS_HIP_BOARD_SLOTS + 2] = {0};
    std::string err;
    uint64_t n_allreduce = 0, n_two_loop = 0, n_two_loop_allreduce = 0;
    bool in_two_loop = false;
    uint64_t n_gram = 0;
    // bus communicators (P2P / RCCL stand-ins)
    std::vector<BusSeg*> targets;   // where this rank's contribution goes: every rank's mailbox (P2P) or the one shared segment (RCCL)
    BusSeg* inbox = nullptr;        // where it reads every rank's contribution
    uint64_t bus_epoch = 0;
    double bus_timeout_s = 20.0;
    bool exclusive = false, gated = false;
    int peers_device = 0, peers_host = 0;
    FILE* plog = nullptr;           // LBFGS_MOCK_PROTOCOL_LOG
};
struct lbfgs_hip_vec {
    lbfgs_hip_ctx* ctx;
    std::vector<double>* p;
};
struct lbfgs_hip_history {
    lbfgs_hip_ctx* ctx;
    int m;
    std::vector<lbfgs_hip_vec*> s, y;
    std::vector<double> ys, alpha;
};

namespace {
std::string g_err;
int fail(lbfgs_hip_ctx* c, int code, const char* msg) {
    (c ? c->err : g_err) = msg;
    return code;
}
bool slot_ok(int first, int count) { return first >= 0 && count >= 0 && first + count <= LBFGS_HIP_BOARD_SLOTS; }
size_t nl(const lbfgs_hip_ctx* c) { return (size_t)c->shard.n_local; }

BusSeg* bus_map(const char* name, bool create, std::string* err) {
    int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) { *err = std::string("shm_open(") + name + "): " + strerror(errno); return nullptr; }
    if (create && ftruncate(fd, (off_t)BUS_BYTES) != 0) { *err = "ftruncate failed"; close(fd); shm_unlink(name); return nullptr; }
    void* p = mmap(nullptr, BUS_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { *err = "mmap failed"; if (create) shm_unlink(name); return nullptr; }
    auto* sgm = new BusSeg();
    sgm->slots = static_cast<BusSlot*>(p);   // (a fresh segment is zero-filled: every epoch 0)
    sgm->name = name;
    sgm->owner = create;
    return sgm;
}
void bus_unmap(BusSeg* sgm) {
    if (!sgm) return;
    munmap(sgm->slots, BUS_BYTES);
    if (sgm->owner) shm_unlink(sgm->name.c_str());
    delete sgm;
}
std::string bus_fresh_name(const char* what) {
    static std::atomic<unsigned> ctr{0};
    char b[56];
    snprintf(b, sizeof(b), "/lbfgs_mock_%s_%d_%u", what, (int)getpid(), ctr.fetch_add(1));
    return b;
}
// in-place sum all-reduce over the bus, in rank order (the same bits on every rank); `tag` names the collective
int bus_allreduce(lbfgs_hip_ctx* c, double* v, int count, uint64_t tag) {
    if (count > BUS_COUNT) return fail(c, LBFGS_HIP_ERR_ARG, "mock bus: too many values in one all-reduce");
    const int W = c->shard.world, me = c->shard.rank;
    const uint64_t e = ++c->bus_epoch;
    tag = (tag << 8) | (uint64_t)count;
    if (c->plog) { fprintf(c->plog, "%llu %llx\n", (unsigned long long)e, (unsigned long long)tag); fflush(c->plog); }
    for (BusSeg* t : c->targets) {
        BusSlot& sl = t->slots[(e % BUS_RING) * BUS_WORLD + me];
        for (int i = 0; i < count; ++i) sl.v[i] = v[i];
        sl.tag = tag;
        sl.epoch.store(e, std::memory_order_release);
    }
    double sum[BUS_COUNT] = {0};
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < W; ++r) {
        BusSlot& sl = c->inbox->slots[(e % BUS_RING) * BUS_WORLD + r];
        while (sl.epoch.load(std::memory_order_acquire) != e) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->bus_timeout_s)
                return fail(c, LBFGS_HIP_ERR_COMM, "mock bus: all-reduce timed out waiting for a peer");
            usleep(50);
        }
        if (sl.tag != tag) return fail(c, LBFGS_HIP_ERR_COMM, "mock bus: the ranks disagree about which collective this is (sequence mismatch)");
        for (int i = 0; i < count; ++i) sum[i] += sl.v[i];
    }
    for (int i = 0; i < count; ++i) v[i] = sum[i];
    return LBFGS_HIP_OK;
}

int allreduce(lbfgs_hip_ctx* c, double* v, int count) {
    if (c->comm_kind == LBFGS_HIP_COMM_NONE) return LBFGS_HIP_OK;
    c->n_allreduce += 1;
    if (c->in_two_loop) c->n_two_loop_allreduce += 1;
    if (c->comm_kind == LBFGS_HIP_COMM_P2P || c->comm_kind == LBFGS_HIP_COMM_RCCL) return bus_allreduce(c, v, count, 0xA11);
    if (c->cb(c->cb_user, v, count) != 0) return fail(c, LBFGS_HIP_ERR_COMM, "all-reduce callback failed");
    return LBFGS_HIP_OK;
}

// the steps of the gated exchange's preparation as the product's collective skeleton sees them (ext_protocol.h), with the
// failures of context.hip's LBFGS_HIP_RESIDENT_FAULT injected on ONE rank (LBFGS_MOCK_FAULT_RANK)
struct MockExtOps {
    lbfgs_hip_ctx* c;
    int fault = 0;
    bool has_comm() const { return true; }
    int setup() { return fault == -2 ? fail(c, LBFGS_HIP_ERR_HIP, "injected: the second stream could not be created (tests)") : 0; }
    int handshake(bool* passed) {
        const char* e = getenv("LBFGS_MOCK_HANDSHAKE_FAIL_RANK");
        *passed = !(e && atoi(e) == c->shard.rank);
        return 0;
    }
    bool shard_ok() { return c->shard.n_local > 0; }  // (an empty shard cannot take the persistent kernel: lbfgs_hip.hip resident_plan)
    int agree(int* bad) {
        double v = *bad ? 1.0 : 0.0;
        const int rc = bus_allreduce(c, &v, 1, 0xA62EE);
        *bad = (int)v;
        return rc;
    }
    int warmup() { double v = 0.0; return bus_allreduce(c, &v, 1, 0x3A63); }
    int after_warmup() { return 0; }
    int trial(bool* ok, bool* enqueued) {
        if (fault == -3) return fail(c, LBFGS_HIP_ERR_HIP, "injected: this rank failed before the gated exchange's trial (tests)");
        double v = (double)(c->shard.rank + 1);
        *enqueued = true;
        const int rc = bus_allreduce(c, &v, 1, 0x721A1);
        const double W = (double)c->shard.world;
        *ok = rc == 0 && v == 0.5 * W * (W + 1.0) && fault != -1;
        return rc;
    }
    int bare_allreduce() { double v = 0.0; return bus_allreduce(c, &v, 1, 0x721A1); }
};
int gdot(lbfgs_hip_ctx* c, const double* a, const double* b, double* out) {
    *out = oracle_vecdot(a, b, nl(c));
    return allreduce(c, out, 1);
}
// clamp a global [start,end) to this shard, in local indices
void local_range(const lbfgs_hip_ctx* c, uint64_t start, uint64_t end, size_t* ls, size_t* le) {
    const uint64_t lo = c->shard.offset, hi = c->shard.offset + c->shard.n_local;
    uint64_t s = start < lo ? lo : start, e = end > hi ? hi : end;
    if (s >= e) { *ls = *le = 0; return; }
    *ls = (size_t)(s - lo);
    *le = (size_t)(e - lo);
}
}  // namespace

extern "C" {

int lbfgs_hip_abi_version(void) { return LBFGS_HIP_ABI_VERSION; }
int lbfgs_hip_device_count(int* count) { *count = 0; return LBFGS_HIP_OK; }
// "RCCL": rank 0 creates the shared segment and its name is the unique id (LBFGS_MOCK_RCCL=1; otherwise the double has none)
int lbfgs_hip_rccl_unique_id(void* id128) {
    if (!getenv("LBFGS_MOCK_RCCL")) return fail(nullptr, LBFGS_HIP_ERR_COMM, "mock: no RCCL");
    const std::string name = bus_fresh_name("rccl");
    std::string err;
    BusSeg* sgm = bus_map(name.c_str(), true, &err);
    if (!sgm) return fail(nullptr, LBFGS_HIP_ERR_COMM, err.c_str());
    sgm->owner = false;  // (the contexts unlink it: the last one to go finds the name gone already, which is fine)
    munmap(sgm->slots, BUS_BYTES);
    delete sgm;
    memset(id128, 0, 128);
    memcpy(id128, BUS_MAGIC, sizeof(BUS_MAGIC));
    snprintf(static_cast<char*>(id128) + 8, 120, "%s", name.c_str());
    return LBFGS_HIP_OK;
}
// "P2P": a mailbox is a segment of this rank's own; the 64-byte handle carries the placement that was ASKED for and the name
int lbfgs_hip_p2p_mailbox_create2(int, int placement, void** mbox, void* handle64) {
    if (getenv("LBFGS_MOCK_NO_P2P")) return fail(nullptr, LBFGS_HIP_ERR_COMM, "mock: no P2P");
    if (placement == LBFGS_HIP_MAILBOX_AUTO) {
        const char* e = getenv("LBFGS_HIP_P2P_MAILBOX");
        placement = (e && strcmp(e, "host") == 0) ? LBFGS_HIP_MAILBOX_HOST : LBFGS_HIP_MAILBOX_DEVICE;
    }
    std::string err;
    BusSeg* sgm = bus_map(bus_fresh_name("mbox").c_str(), true, &err);
    if (!sgm) return fail(nullptr, LBFGS_HIP_ERR_COMM, err.c_str());
    sgm->placement = placement;
    memset(handle64, 0, 64);
    memcpy(handle64, BUS_MAGIC, sizeof(BUS_MAGIC));
    static_cast<char*>(handle64)[8] = (char)placement;
    snprintf(static_cast<char*>(handle64) + 9, 55, "%s", sgm->name.c_str());
    *mbox = sgm;
    return LBFGS_HIP_OK;
}
int lbfgs_hip_p2p_mailbox_create(int dev, void** mbox, void* handle64) { return lbfgs_hip_p2p_mailbox_create2(dev, LBFGS_HIP_MAILBOX_AUTO, mbox, handle64); }
void lbfgs_hip_p2p_mailbox_destroy(int, void* mbox) { bus_unmap(static_cast<BusSeg*>(mbox)); }
int lbfgs_hip_ctx_p2p_seal(lbfgs_hip_ctx* ctx, int* placement_out) {
    if (!ctx) return LBFGS_HIP_ERR_ARG;
    if (placement_out) *placement_out = (ctx->comm_kind == LBFGS_HIP_COMM_P2P && ctx->inbox) ? ctx->inbox->placement : -1;
    return LBFGS_HIP_OK;
}
#ifndef LBFGS_MOCK_BUILD_ID
#define LBFGS_MOCK_BUILD_ID "unstamped"
#endif
static const char mock_build_id_marker[] = "LBFGS_MOCK_BUILD_ID=" LBFGS_MOCK_BUILD_ID;
const char* lbfgs_hip_build_id(void) { return mock_build_id_marker + sizeof("LBFGS_MOCK_BUILD_ID=") - 1; }

int lbfgs_hip_ctx_create(lbfgs_hip_ctx** out, int, uint64_t n, const lbfgs_hip_shard* shard, const lbfgs_hip_comm* comm,
                         void*) {
    auto* c = new lbfgs_hip_ctx();
    if (shard) c->shard = *shard;
    else c->shard = {0, 1, n, 0, n};
    int kind = comm ? comm->kind : LBFGS_HIP_COMM_NONE;
    if (c->shard.world > 1 && kind == LBFGS_HIP_COMM_NONE) { delete c; return fail(nullptr, LBFGS_HIP_ERR_ARG, "world > 1 needs a communicator"); }
    if (c->shard.world > BUS_WORLD && (kind == LBFGS_HIP_COMM_RCCL || kind == LBFGS_HIP_COMM_P2P)) { delete c; return fail(nullptr, LBFGS_HIP_ERR_ARG, "mock bus: world too large"); }
    if (kind == LBFGS_HIP_COMM_CALLBACK) { c->comm_kind = kind; c->cb = comm->callback; c->cb_user = comm->callback_user; }
    if (const char* e = getenv("LBFGS_MOCK_PROTOCOL_LOG")) {
        const std::string path = std::string(e) + ".rank" + std::to_string(c->shard.rank);
        c->plog = fopen(path.c_str(), "a");
    }
    if (kind == LBFGS_HIP_COMM_P2P) {
        // (the product takes ownership of the mailbox whether creation succeeds or not)
        BusSeg* own = static_cast<BusSeg*>(comm->p2p_mailbox);
        if (!own || !comm->p2p_handles) { bus_unmap(own); lbfgs_hip_ctx_destroy(c); return fail(nullptr, LBFGS_HIP_ERR_ARG, "P2P communicator needs a mailbox and handles"); }
        c->comm_kind = kind;
        c->inbox = own;
        c->exclusive = comm->exclusive_device != 0;
        if (comm->p2p_timeout_s > 0) c->bus_timeout_s = comm->p2p_timeout_s;
        c->targets.assign((size_t)c->shard.world, nullptr);
        c->targets[(size_t)c->shard.rank] = own;
        for (int r = 0; r < c->shard.world; ++r) {
            if (r == c->shard.rank) continue;
            const char* h = static_cast<const char*>(comm->p2p_handles) + 64 * r;
            const int placement = (int)h[8];
            std::string err;
            BusSeg* peer = nullptr;
            if (memcmp(h, BUS_MAGIC, sizeof(BUS_MAGIC)) != 0) err = "not a mailbox handle";
            else if (placement == LBFGS_HIP_MAILBOX_DEVICE && getenv("LBFGS_MOCK_NO_DEVICE_IPC"))
                err = "hipIpcOpenMemHandle: invalid argument (mock: this rank sees only its own device)";
            else peer = bus_map(h + 9, false, &err);
            if (!peer) {
                const std::string msg = "mailbox of rank " + std::to_string(r) + ": " + err;
                lbfgs_hip_ctx_destroy(c);
                return fail(nullptr, LBFGS_HIP_ERR_COMM, msg.c_str());
            }
            peer->placement = placement;
            (placement == LBFGS_HIP_MAILBOX_HOST ? c->peers_host : c->peers_device) += 1;
            c->targets[(size_t)r] = peer;
        }
    }
    if (kind == LBFGS_HIP_COMM_RCCL) {
        const char* id = static_cast<const char*>(comm->rccl_unique_id);
        std::string err;
        BusSeg* shared = (id && memcmp(id, BUS_MAGIC, sizeof(BUS_MAGIC)) == 0) ? bus_map(id + 8, false, &err) : nullptr;
        if (!shared) { lbfgs_hip_ctx_destroy(c); return fail(nullptr, LBFGS_HIP_ERR_COMM, ("mock RCCL: " + (err.empty() ? std::string("no unique id") : err)).c_str()); }
        shared->owner = true;  // (every rank tries to unlink the name when it goes)
        c->comm_kind = kind;
        c->inbox = shared;
        c->targets.assign(1, shared);
        c->exclusive = comm->exclusive_device != 0;
        // the gated exchange: opt-in, exclusive devices only, prepared COLLECTIVELY by the product's own skeleton
        const char* opt = getenv("LBFGS_HIP_RCCL_RESIDENT");
        if (c->exclusive && opt && atoi(opt) != 0) {
            MockExtOps ops{c};
            const char* f = getenv("LBFGS_HIP_RESIDENT_FAULT");
            const char* fr = getenv("LBFGS_MOCK_FAULT_RANK");
            if (f && (!fr || atoi(fr) == c->shard.rank)) ops.fault = atoi(f);
            lh::ExtOutcome res;
            const int rc = lh::ext_prepare_protocol(ops, &res);
            if (rc < 0) {
                const std::string msg = c->err;
                lbfgs_hip_ctx_destroy(c);
                return fail(nullptr, rc, msg.c_str());
            }
            c->gated = rc == 0;
            if (rc == 1) fprintf(stderr, "[mock lbfgs_hip] the gated RCCL exchange is not used: stage %d, %d of %d ranks against; kernel per step\n",
                                 res.stage, res.bad, c->shard.world);
        }
    }
    *out = c;
    return LBFGS_HIP_OK;
}
void lbfgs_hip_ctx_destroy(lbfgs_hip_ctx* c) {
    if (!c) return;
    if (c->comm_kind == LBFGS_HIP_COMM_P2P) for (BusSeg* t : c->targets) bus_unmap(t);
    else if (c->inbox) bus_unmap(c->inbox);
    if (c->plog) fclose(c->plog);
    delete c;
}
const char* lbfgs_hip_last_error(const lbfgs_hip_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }
int lbfgs_hip_sync(lbfgs_hip_ctx*) { return LBFGS_HIP_OK; }
void* lbfgs_hip_stream(lbfgs_hip_ctx*) { return nullptr; }
int lbfgs_hip_get_shard(const lbfgs_hip_ctx* c, lbfgs_hip_shard* out) { *out = c->shard; return LBFGS_HIP_OK; }
int lbfgs_hip_set_grid(lbfgs_hip_ctx*, int) { return LBFGS_HIP_OK; }

int lbfgs_hip_vec_alloc(lbfgs_hip_ctx* c, lbfgs_hip_vec** out) {
    *out = new lbfgs_hip_vec{c, new std::vector<double>(nl(c) ? nl(c) : 1, 0.0)};
    return LBFGS_HIP_OK;
}
void lbfgs_hip_vec_free(lbfgs_hip_vec* v) { if (v) { delete v->p; delete v; } }
int lbfgs_hip_vec_upload(lbfgs_hip_vec* v, const double* h, uint64_t n) {
    if (n != v->ctx->shard.n_local) return fail(v->ctx, LBFGS_HIP_ERR_ARG, "size mismatch");
    memcpy(v->p->data(), h, n * sizeof(double));
    return LBFGS_HIP_OK;
}
int lbfgs_hip_vec_download(const lbfgs_hip_vec* v, double* h, uint64_t n) {
    if (n != v->ctx->shard.n_local) return fail(v->ctx, LBFGS_HIP_ERR_ARG, "size mismatch");
    memcpy(h, v->p->data(), n * sizeof(double));
    return LBFGS_HIP_OK;
}
int lbfgs_hip_vec_fill(lbfgs_hip_vec* v, double c) { for (auto& e : *v->p) e = c; return LBFGS_HIP_OK; }
void* lbfgs_hip_vec_ptr(lbfgs_hip_vec* v) { return v->p->data(); }
int lbfgs_hip_vec_swap(lbfgs_hip_vec* a, lbfgs_hip_vec* b) { std::swap(a->p, b->p); return LBFGS_HIP_OK; }

int lbfgs_hip_scalars_read(lbfgs_hip_ctx* c, int f, int n, double* h) {
    if (!slot_ok(f, n)) return LBFGS_HIP_ERR_ARG;
    memcpy(h, c->board + f, n * sizeof(double));
    return LBFGS_HIP_OK;
}
int lbfgs_hip_scalars_write(lbfgs_hip_ctx* c, int f, int n, const double* h) {
    if (!slot_ok(f, n)) return LBFGS_HIP_ERR_ARG;
    memcpy(c->board + f, h, n * sizeof(double));
    return LBFGS_HIP_OK;
}
void* lbfgs_hip_scalars_ptr(lbfgs_hip_ctx* c) { return c->board; }
int lbfgs_hip_scalars_allreduce(lbfgs_hip_ctx* c, int f, int n) {
    if (!slot_ok(f, n)) return LBFGS_HIP_ERR_ARG;
    return allreduce(c, c->board + f, n);
}

int lbfgs_hip_vecadd(lbfgs_hip_vec* y, const lbfgs_hip_vec* x, double c) { oracle_vecadd(y->p->data(), x->p->data(), c, nl(y->ctx)); return 0; }
int lbfgs_hip_vecadd_dev(lbfgs_hip_vec* y, const lbfgs_hip_vec* x, int s) { return lbfgs_hip_vecadd(y, x, y->ctx->board[s]); }
int lbfgs_hip_vecdot(const lbfgs_hip_vec* x, const lbfgs_hip_vec* y, int s) { return gdot(x->ctx, x->p->data(), y->p->data(), x->ctx->board + s); }
int lbfgs_hip_vecscale(lbfgs_hip_vec* y, double c) { oracle_vecscale(y->p->data(), c, nl(y->ctx)); return 0; }
int lbfgs_hip_veccpy(lbfgs_hip_vec* y, const lbfgs_hip_vec* x) { oracle_veccpy(y->p->data(), x->p->data(), nl(y->ctx)); return 0; }
int lbfgs_hip_vecncpy(lbfgs_hip_vec* y, const lbfgs_hip_vec* x) { oracle_vecncpy(y->p->data(), x->p->data(), nl(y->ctx)); return 0; }
int lbfgs_hip_vecdiff(lbfgs_hip_vec* z, const lbfgs_hip_vec* x, const lbfgs_hip_vec* y) { oracle_vecdiff(z->p->data(), x->p->data(), y->p->data(), nl(z->ctx)); return 0; }
int lbfgs_hip_vec2norm_sq(const lbfgs_hip_vec* x, int s) { return gdot(x->ctx, x->p->data(), x->p->data(), x->ctx->board + s); }
int lbfgs_hip_vec2norm(const lbfgs_hip_vec* x, int s, double* out) {
    int rc = lbfgs_hip_vec2norm_sq(x, s);
    if (rc == 0) *out = std::sqrt(x->ctx->board[s]);
    return rc;
}
int lbfgs_hip_vec2norminv(const lbfgs_hip_vec* x, int s, double* out) {
    double v = 0.0;
    int rc = lbfgs_hip_vec2norm(x, s, &v);
    if (rc == 0) *out = 1.0 / v;
    return rc;
}

int lbfgs_hip_line_step(lbfgs_hip_vec* x, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double step,
                        const lbfgs_hip_vec* wp, uint64_t start, uint64_t end) {
    lbfgs_hip_ctx* c = x->ctx;
    oracle_veccpy(x->p->data(), xp->p->data(), nl(c));
    oracle_vecadd(x->p->data(), d->p->data(), step, nl(c));
    if (wp) {
        size_t ls, le;
        local_range(c, start, end, &ls, &le);
        oracle_project(x->p->data(), wp->p->data(), ls, le, 0);
    }
    return LBFGS_HIP_OK;
}
int lbfgs_hip_norms_sq(const lbfgs_hip_vec* x, const lbfgs_hip_vec* g, int s) {
    lbfgs_hip_ctx* c = x->ctx;
    c->board[s] = oracle_vecdot(x->p->data(), x->p->data(), nl(c));
    c->board[s + 1] = oracle_vecdot(g->p->data(), g->p->data(), nl(c));
    return allreduce(c, c->board + s, 2);
}

int lbfgs_hip_history_create(lbfgs_hip_ctx* c, int m, lbfgs_hip_history** out) {
    auto* h = new lbfgs_hip_history{c, m, {}, {}, std::vector<double>(m, 0.0), std::vector<double>(m, 0.0)};
    for (int i = 0; i < m; ++i) {
        lbfgs_hip_vec *s, *y;
        lbfgs_hip_vec_alloc(c, &s);
        lbfgs_hip_vec_alloc(c, &y);
        h->s.push_back(s);
        h->y.push_back(y);
    }
    *out = h;
    return LBFGS_HIP_OK;
}
void lbfgs_hip_history_destroy(lbfgs_hip_history* h) {
    if (!h) return;
    for (auto* v : h->s) lbfgs_hip_vec_free(v);
    for (auto* v : h->y) lbfgs_hip_vec_free(v);
    delete h;
}
lbfgs_hip_vec* lbfgs_hip_history_s(lbfgs_hip_history* h, int i) { return (i >= 0 && i < h->m) ? h->s[i] : nullptr; }
lbfgs_hip_vec* lbfgs_hip_history_y(lbfgs_hip_history* h, int i) { return (i >= 0 && i < h->m) ? h->y[i] : nullptr; }
int lbfgs_hip_history_scalars_read(lbfgs_hip_history* h, double* ys, double* alpha) {
    if (ys) memcpy(ys, h->ys.data(), h->m * sizeof(double));
    if (alpha) memcpy(alpha, h->alpha.data(), h->m * sizeof(double));
    return LBFGS_HIP_OK;
}
int lbfgs_hip_history_scalars_write(lbfgs_hip_history* h, const double* ys, const double* alpha) {
    if (ys) memcpy(h->ys.data(), ys, h->m * sizeof(double));
    if (alpha) memcpy(h->alpha.data(), alpha, h->m * sizeof(double));
    return LBFGS_HIP_OK;
}

int lbfgs_hip_history_update(lbfgs_hip_history* h, int slot, const lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                             const lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, double step, int damping, int o) {
    lbfgs_hip_ctx* c = h->ctx;
    const size_t n = nl(c);
    double* s = h->s[slot]->p->data();
    double* y = h->y[slot]->p->data();
    oracle_vecdiff(s, x->p->data(), xp->p->data(), n);
    oracle_vecdiff(y, g->p->data(), gp->p->data(), n);
    double* b = c->board + o;
    b[0] = oracle_vecdot(s, s, n);
    b[1] = oracle_vecdot(y, s, n);
    b[2] = oracle_vecdot(y, y, n);
    b[3] = oracle_vecdot(x->p->data(), x->p->data(), n);
    b[4] = oracle_vecdot(g->p->data(), g->p->data(), n);
    b[5] = 0.0;
    if (damping) {
        std::vector<double> bs(gp->p->begin(), gp->p->begin() + n);
        oracle_vecscale(bs.data(), -step, n);
        b[5] = oracle_vecdot(s, bs.data(), n);
    }
    {   // s.(-g) as the reference's first two-loop dot computes it: vecdot(s, d) with d = -g
        std::vector<double> d(n);
        oracle_vecncpy(d.data(), g->p->data(), n);
        b[6] = oracle_vecdot(s, d.data(), n);
    }
    int rc = allreduce(c, b, 7);
    h->ys[slot] = b[1];
    return rc;
}
int lbfgs_hip_history_damp(lbfgs_hip_history* h, int slot, const lbfgs_hip_vec* gp, double step, double theta) {
    const size_t n = nl(h->ctx);
    std::vector<double> bs(gp->p->begin(), gp->p->begin() + n);
    oracle_vecscale(bs.data(), -step, n);
    oracle_vecscale(bs.data(), 1.0 - theta, n);
    oracle_vecadd(bs.data(), h->y[slot]->p->data(), theta, n);
    oracle_veccpy(h->y[slot]->p->data(), bs.data(), n);
    return LBFGS_HIP_OK;
}

int lbfgs_hip_two_loop_unfused(lbfgs_hip_history* h, lbfgs_hip_vec* d, uint64_t k, int end, int gn, int gd,
                               int* new_end) {
    lbfgs_hip_ctx* c = h->ctx;
    const size_t n = nl(c);
    const int m = h->m;
    const int e1 = (end + 1) % m;
    const int bound = (int)((uint64_t)m < k ? (uint64_t)m : k);
    double* dd = d->p->data();
    int j = e1, rc;
    for (int it = 0; it < bound; ++it) {
        j = (j + m - 1) % m;
        double dot;
        if ((rc = gdot(c, h->s[j]->p->data(), dd, &dot)) != 0) return rc;
        h->alpha[j] = dot / h->ys[j];
        oracle_vecadd(dd, h->y[j]->p->data(), -h->alpha[j], n);
    }
    oracle_vecscale(dd, c->board[gn] / c->board[gd], n);
    for (int it = 0; it < bound; ++it) {
        double dot;
        if ((rc = gdot(c, h->y[j]->p->data(), dd, &dot)) != 0) return rc;
        const double beta = dot / h->ys[j];
        oracle_vecadd(dd, h->s[j]->p->data(), h->alpha[j] - beta, n);
        j = (j + 1) % m;
    }
    *new_end = e1;
    return LBFGS_HIP_OK;
}
int lbfgs_hip_two_loop(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end, int gn,
                       int gd, int dn, int* new_end) {
    struct Scope {  // (lbfgs_hip_ctx_comm_info: all-reduces inside a two-loop are counted apart)
        lbfgs_hip_ctx* c;
        explicit Scope(lbfgs_hip_ctx* cc) : c(cc) { c->n_two_loop += 1; c->in_two_loop = true; }
        ~Scope() { c->in_two_loop = false; }
    } scope(h->ctx);
    oracle_vecncpy(d->p->data(), g->p->data(), nl(h->ctx));
    int rc = lbfgs_hip_two_loop_unfused(h, d, k, end, gn, gd, new_end);
    if (rc != 0) return rc;
    rc = gdot(h->ctx, d->p->data(), d->p->data(), h->ctx->board + dn);
    if (rc != 0) return rc;
    return gdot(h->ctx, g->p->data(), d->p->data(), h->ctx->board + dn + 1);
}

int lbfgs_hip_two_loop_from(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end, int gn,
                            int gd, int dn, int, int* new_end) {
    return lbfgs_hip_two_loop(h, d, g, k, end, gn, gd, dn, new_end);  // the test double always recomputes the dot
}

int lbfgs_hip_constrain_direction(lbfgs_hip_vec* d, const lbfgs_hip_vec* pg, uint64_t start, uint64_t end, int o);
int lbfgs_hip_two_loop_owlqn(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* pg, uint64_t k, int end, int gn,
                             int gd, int dn, uint64_t start, uint64_t end_, int* new_end) {
    int rc = lbfgs_hip_two_loop(h, d, pg, k, end, gn, gd, dn, new_end);
    if (rc != 0) return rc;
    return lbfgs_hip_constrain_direction(d, pg, start, end_, dn + 2);
}

int lbfgs_hip_two_loop_gram(lbfgs_hip_history* h, lbfgs_hip_vec* d, const lbfgs_hip_vec* g, uint64_t k, int end, int gn,
                            int gd, int dn, int* new_end) {
    const int rc = lbfgs_hip_two_loop(h, d, g, k, end, gn, gd, dn, new_end);  // the test double has one recursion
    // the coefficient-space prediction of ||d||^2 (include/lbfgs_hip.h): exact here -- unless a test asks for a direction
    // that fails its check (LBFGS_MOCK_VF_BAD = every k-th call, counted per context)
    h->ctx->board[dn + 2] = h->ctx->board[dn];
    h->ctx->board[dn + 3] = 1.0;  // (no cancellation to report: the test double runs the exact recursion)
    if (const char* e = getenv("LBFGS_MOCK_VF_BAD")) {
        const int every = atoi(e);
        h->ctx->n_gram += 1;
        if (every > 0 && h->ctx->n_gram % (uint64_t)every == 0) h->ctx->board[dn + 2] = 2.0 * h->ctx->board[dn] + 1.0;
    }
    return rc;
}

int lbfgs_hip_owlqn_post_eval(const lbfgs_hip_vec* x, const lbfgs_hip_vec* g, lbfgs_hip_vec* pg, double cc,
                              uint64_t start, uint64_t end, int o) {
    lbfgs_hip_ctx* c = x->ctx;
    size_t ls, le;
    local_range(c, start, end, &ls, &le);
    double* b = c->board + o;
    b[0] = oracle_x1norm(cc, ls, le, x->p->data());
    oracle_pseudo_gradient(cc, ls, le, pg->p->data(), x->p->data(), g->p->data(), nl(c));
    b[1] = oracle_vecdot(pg->p->data(), pg->p->data(), nl(c));
    b[2] = oracle_vecdot(x->p->data(), x->p->data(), nl(c));
    return allreduce(c, b, 3);
}
int lbfgs_hip_orthant_select(lbfgs_hip_vec* wp, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* pg) {
    oracle_orthant_select(wp->p->data(), xp->p->data(), pg->p->data(), nl(wp->ctx));
    return LBFGS_HIP_OK;
}
int lbfgs_hip_constrain_direction(lbfgs_hip_vec* d, const lbfgs_hip_vec* pg, uint64_t start, uint64_t end, int o) {
    size_t ls, le;
    local_range(d->ctx, start, end, &ls, &le);
    oracle_project(d->p->data(), pg->p->data(), ls, le, 1);
    int rc = gdot(d->ctx, d->p->data(), d->p->data(), d->ctx->board + o);
    if (rc != 0) return rc;
    return gdot(d->ctx, pg->p->data(), d->p->data(), d->ctx->board + o + 1);
}

static int eval_obj(const lbfgs_hip_objective* obj, lbfgs_hip_ctx* c, const double* x, double* g, double* f) {
    oracle_hashed_obj u{c->shard.offset, obj->seed_a, obj->seed_b};
    int failed = 0;
    switch (obj->kind) {
        case LBFGS_HIP_OBJ_QUADRATIC: *f = oracle_obj_quadratic(&u, x, g, nl(c), &failed); break;
        case LBFGS_HIP_OBJ_LOGISTIC: *f = oracle_obj_logistic(&u, x, g, nl(c), &failed); break;
        case LBFGS_HIP_OBJ_ROSENBROCK:
            if ((c->shard.n_local & 1) || (c->shard.offset & 1)) return fail(c, LBFGS_HIP_ERR_ARG, "Rosenbrock needs even shards");
            *f = oracle_obj_rosenbrock(nullptr, x, g, nl(c), &failed);
            break;
        case LBFGS_HIP_OBJ_LJ_ALLPAIRS: *f = oracle_obj_lj(nullptr, x, g, nl(c), &failed); break;
        case LBFGS_HIP_OBJ_LJ_NEIGHBORS: {
            oracle_lj_nbr u2{(const int32_t*)obj->nbr_index, obj->max_nbr, 0, obj->cutoff};
            *f = oracle_obj_lj_neighbors(&u2, x, g, nl(c), &failed);
            break;
        }
        case LBFGS_HIP_OBJ_LJ_CELLS: {  // the cutoff rule itself (no list to carry on the CPU)
            oracle_lj_cells u3{obj->cutoff};
            *f = oracle_obj_lj_cells(&u3, x, g, nl(c), &failed);
            if (failed) return fail(c, LBFGS_HIP_ERR_ARG, "LJ_CELLS: a coordinate is not finite");
            break;
        }
        default: return fail(c, LBFGS_HIP_ERR_ARG, "unknown objective");
    }
    return LBFGS_HIP_OK;
}
int lbfgs_hip_path_stats(lbfgs_hip_ctx*, uint64_t* r, uint64_t* e) { if (r) *r = 0; if (e) *e = 0; return LBFGS_HIP_OK; }
int lbfgs_hip_ctx_comm_info(lbfgs_hip_ctx* c, lbfgs_hip_comm_info* out) {
    if (!c || !out) return LBFGS_HIP_ERR_ARG;
    memset(out, 0, sizeof(*out));
    out->kind = c->comm_kind;
    out->world = c->shard.world;
    out->rank = c->shard.rank;
    out->ranks_seen = c->comm_kind == LBFGS_HIP_COMM_NONE ? 1 : 0;  // (callback: the library sees no peer itself)
    out->rank_seen = c->shard.rank;
    out->mailbox_placement = -1;
    out->exclusive_device = c->exclusive ? 1 : 0;
    if (c->comm_kind == LBFGS_HIP_COMM_RCCL) out->ranks_seen = c->shard.world;
    if (c->comm_kind == LBFGS_HIP_COMM_P2P) {
        out->ranks_seen = 1 + c->peers_device + c->peers_host;
        out->mailbox_placement = c->inbox->placement;
        out->peers_device = c->peers_device;
        out->peers_host = c->peers_host;
        out->p2p_exchanges = c->n_allreduce;
    }
    out->two_loops = c->n_two_loop;
    out->two_loop_exchanges = c->n_two_loop_allreduce;
    out->allreduce_launches = c->n_allreduce;
    return LBFGS_HIP_OK;
}
int lbfgs_hip_lj_cells_stats(lbfgs_hip_ctx*, uint64_t* rebuilds, uint64_t* evaluations, uint32_t* longest) {
    if (rebuilds) *rebuilds = 0;
    if (evaluations) *evaluations = 0;
    if (longest) *longest = 0;
    return LBFGS_HIP_OK;
}
int lbfgs_hip_host_buffer_create(lbfgs_hip_ctx*, uint64_t bytes, void** out) {
    *out = malloc(bytes ? bytes : 1);
    return *out ? LBFGS_HIP_OK : LBFGS_HIP_ERR_NOMEM;
}
void lbfgs_hip_host_buffer_destroy(lbfgs_hip_ctx*, void* buf) { free(buf); }
int lbfgs_hip_device_buffer_create(lbfgs_hip_ctx*, const void* host, uint64_t bytes, void** out) {
    *out = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(*out, host, bytes);
    return LBFGS_HIP_OK;
}
void lbfgs_hip_device_buffer_destroy(lbfgs_hip_ctx*, void* buf) { free(buf); }
int lbfgs_hip_objective_eval(const lbfgs_hip_objective* obj, const lbfgs_hip_vec* x, lbfgs_hip_vec* g, int o) {
    lbfgs_hip_ctx* c = x->ctx;
    int rc = eval_obj(obj, c, x->p->data(), g->p->data(), c->board + o);
    return rc ? rc : allreduce(c, c->board + o, 1);
}
int lbfgs_hip_objective_line_eval(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                  const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* g, int o) {
    lbfgs_hip_ctx* c = x->ctx;
    lbfgs_hip_line_step(x, xp, d, step, nullptr, 0, 0);
    int rc = eval_obj(obj, c, x->p->data(), g->p->data(), c->board + o);
    if (rc) return rc;
    c->board[o + 1] = oracle_vecdot(g->p->data(), d->p->data(), nl(c));
    return allreduce(c, c->board + o, 2);
}

int lbfgs_hip_objective_is_elementwise(const lbfgs_hip_objective* obj) {
    return obj && (obj->kind == LBFGS_HIP_OBJ_QUADRATIC || obj->kind == LBFGS_HIP_OBJ_LOGISTIC);
}
// the deferred-trial pair as the reference's own sequence on scratch / on the real vectors
int lbfgs_hip_objective_line_probe(const lbfgs_hip_objective* obj, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d,
                                   double step, int o) {
    lbfgs_hip_ctx* c = xp->ctx;
    if (!lbfgs_hip_objective_is_elementwise(obj)) return fail(c, LBFGS_HIP_ERR_ARG, "not elementwise");
    const size_t n = nl(c);
    std::vector<double> x(n), g(n);
    oracle_veccpy(x.data(), xp->p->data(), n);
    oracle_vecadd(x.data(), d->p->data(), step, n);
    int rc = eval_obj(obj, c, x.data(), g.data(), c->board + o);
    if (rc) return rc;
    c->board[o + 1] = oracle_vecdot(g.data(), d->p->data(), n);
    return allreduce(c, c->board + o, 2);
}
int lbfgs_hip_history_update_from_step(lbfgs_hip_history* h, int slot, const lbfgs_hip_objective* obj,
                                       lbfgs_hip_vec* x, const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double t,
                                       lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, double step, int damping, int o) {
    lbfgs_hip_ctx* c = h->ctx;
    if (!lbfgs_hip_objective_is_elementwise(obj)) return fail(c, LBFGS_HIP_ERR_ARG, "not elementwise");
    lbfgs_hip_line_step(x, xp, d, t, nullptr, 0, 0);
    double f;
    int rc = eval_obj(obj, c, x->p->data(), g->p->data(), &f);
    if (rc) return rc;
    return lbfgs_hip_history_update(h, slot, x, xp, g, gp, step, damping, o);
}

int lbfgs_hip_objective_owlqn_line_eval(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                        const lbfgs_hip_vec* d, double step, const lbfgs_hip_vec* wp, lbfgs_hip_vec* g,
                                        lbfgs_hip_vec* pg, double c, uint64_t start, uint64_t end, int o) {
    int rc = lbfgs_hip_line_step(x, xp, d, step, wp, start, end);
    if (rc) return rc;
    if ((rc = lbfgs_hip_objective_eval(obj, x, g, o))) return rc;
    if ((rc = lbfgs_hip_owlqn_post_eval(x, g, pg, c, start, end, o + 2))) return rc;
    return lbfgs_hip_vecdot(g, d, o + 1);
}

int lbfgs_hip_objective_owlqn_first_trial(const lbfgs_hip_objective* obj, lbfgs_hip_vec* x, const lbfgs_hip_vec* xp,
                                          const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* wp, lbfgs_hip_vec* g,
                                          lbfgs_hip_vec* pg, double c, uint64_t start, uint64_t end, int o) {
    int rc = lbfgs_hip_orthant_select(wp, xp, pg);  // core.rs:167-180, then the trial (line.rs:735, :740-743)
    if (rc) return rc;
    return lbfgs_hip_objective_owlqn_line_eval(obj, x, xp, d, step, wp, g, pg, c, start, end, o);
}

int lbfgs_hip_objective_owlqn_trial_update(const lbfgs_hip_objective* obj, lbfgs_hip_history* h, int slot, lbfgs_hip_vec* x,
                                           const lbfgs_hip_vec* xp, const lbfgs_hip_vec* d, double step, lbfgs_hip_vec* wp,
                                           int first, lbfgs_hip_vec* g, const lbfgs_hip_vec* gp, lbfgs_hip_vec* pg, double c,
                                           uint64_t start, uint64_t end, int o, int upd) {
    // the trial, then IterationData::update for its point (lbfgs.rs:640-656): the unfused sequence, bit for bit
    int rc = first ? lbfgs_hip_objective_owlqn_first_trial(obj, x, xp, d, step, wp, g, pg, c, start, end, o)
                   : lbfgs_hip_objective_owlqn_line_eval(obj, x, xp, d, step, wp, g, pg, c, start, end, o);
    if (rc) return rc;
    return lbfgs_hip_history_update(h, slot, x, xp, g, gp, step, 0, upd);
}

int lbfgs_hip_prof_enable(lbfgs_hip_ctx*, int) { return LBFGS_HIP_OK; }
int lbfgs_hip_prof_reset(lbfgs_hip_ctx*) { return LBFGS_HIP_OK; }
int lbfgs_hip_prof_read(lbfgs_hip_ctx* c, int k, uint64_t* launches, double* ms) {
    // the mock reports its all-reduce count through the COMM class (used by the sharding tests)
    if (launches) *launches = (k == LBFGS_HIP_K_COMM) ? c->n_allreduce : 0;
    if (ms) *ms = 0.0;
    // LBFGS_MOCK_FAKE_KERNEL_TIMES=1: one two-loop step "timed" at 1 ms, so that bench.py's record has a dominant kernel and
    // the orchestration that follows its measurement (counter passes, budget, signals) runs on the test double
    static const bool fake = getenv("LBFGS_MOCK_FAKE_KERNEL_TIMES") != nullptr;
    if (fake && (k == LBFGS_HIP_K_TWOLOOP_STEP || k == LBFGS_HIP_K_TWOLOOP_ALL)) {
        if (launches) *launches = 1;
        if (ms) *ms = 1.0;
    }
    // ... and the launch FORM the product would have taken: the persistent kernel for one rank, for P2P ranks that own their GPU
    // and under RCCL only with the gated exchange agreed on by every rank; LBFGS_HIP_RESIDENT=0 turns it off (bench.ran_as)
    const char* res = getenv("LBFGS_HIP_RESIDENT");
    const bool resident = !(res && atoi(res) == 0) && (c->comm_kind == LBFGS_HIP_COMM_NONE || (c->comm_kind == LBFGS_HIP_COMM_P2P && c->exclusive) ||
                                                        (c->comm_kind == LBFGS_HIP_COMM_RCCL && c->gated));
    if (fake && k == LBFGS_HIP_K_TWOLOOP_RESIDENT && resident) {
        if (launches) *launches = 1;
        if (ms) *ms = 1.0;
    }
    return LBFGS_HIP_OK;
}

}  // extern "C"
