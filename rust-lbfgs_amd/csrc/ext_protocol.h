// rust-lbfgs_amd/csrc/ext_protocol.h -- the COLLECTIVE skeleton of preparing the gated RCCL exchange (context.hip ext_prepare).
//
// lbfgs_hip_ctx_create decides, together with every other rank of the communicator, whether the two-loop recursion may run as
// ONE persistent kernel whose hand-offs are closed by ncclAllReduce launches gated on a second stream (stream.h ext_exchange).
// Every rank must end with the SAME answer -- a rank with a kernel per step next to ranks with gated chains would pair its
// all-reduces wrongly -- and every rank must make the SAME sequence of collectives whatever happens to it locally: a rank that
// returned early on a local failure would leave its peers inside ncclAllReduce + hipStreamSynchronize for ever (the host side
// has no timeout).  That sequence is stated once, here, free of HIP, so that the product (context.hip: the HIP steps) and the
// CPU test double (tests/support/mock_lbfgs_hip.cpp: injected failures, several ranks -- which no single GPU can host) run the
// same code.
//
//   Ops (all return 0 or an error code < 0 unless said otherwise):
//     bool has_comm()                   the context has a communicator (false: nothing collective is done)
//     int  setup()                      LOCAL: second stream, uncached block, event
//     int  handshake(bool* passed)      LOCAL: both streams are served concurrently
//     bool shard_ok()                   LOCAL: this rank's shard and device can take the persistent kernel at all
//     int  agree(int* bad)              COLLECTIVE: *bad (0 / 1) -> the number of ranks that said 1
//     int  warmup()                     COLLECTIVE: RCCL's first all-reduce on the second stream
//     int  after_warmup()               LOCAL: wait for it, clear the slot
//     int  trial(bool* ok, bool* enqueued)  the exchange once under a chip-filling kernel; *enqueued: this rank's all-reduce of
//                                       the trial has been handed to the communicator (its peers wait for it)
//     int  bare_allreduce()             COLLECTIVE: the trial's all-reduce without the trial (a rank that could not run it)
//   An error from agree / warmup / bare_allreduce means the communicator itself failed: nothing collective can follow, returned
//   at once.  Everything else is recorded and returned after the last collective.
#pragma once

namespace lh {

struct ExtOutcome {
    int stage = 0;  // where the answer was decided: 0 = a shard or device cannot take the kernel, 1 = setup / handshake, 2 = the trial
    int bad = 0;    // ranks that voted against at that stage (0: the gated exchange is usable on every rank)
};

// -> 0: usable on every rank; 1: not (every rank takes the kernel-per-step form; *out says why); < 0: this rank's error
template <class Ops>
int ext_prepare_protocol(Ops& o, ExtOutcome* out) {
    int local_rc = o.setup();
    bool passed = false;
    if (local_rc == 0) local_rc = o.handshake(&passed);
    const bool shard_ok = o.shard_ok();
    int bad = (local_rc == 0 && passed && shard_ok) ? 0 : 1;
    int stage = (shard_ok || local_rc != 0) ? 1 : 0;
    if (o.has_comm()) {
        const int rc = o.agree(&bad);
        if (rc != 0) return rc;
    }
    if (bad == 0 && o.has_comm()) {  // (the AGREED value: every rank comes here or none does)
        int rc = o.warmup();
        if (rc != 0) return rc;
        local_rc = o.after_warmup();
        bool ok = false, enqueued = false;
        if (local_rc == 0) {
            const int rc_t = o.trial(&ok, &enqueued);
            if (rc_t != 0) local_rc = rc_t;
        }
        if (!enqueued) {  // the peers wait inside RCCL for this rank's all-reduce of the trial: hand them one (the sum is then
            rc = o.bare_allreduce();  // wrong and every rank sees the trial fail -- nobody is left waiting)
            if (rc != 0) return rc;
        }
        bad = (ok && local_rc == 0) ? 0 : 1;
        stage = 2;
        rc = o.agree(&bad);
        if (rc != 0) return rc;
    }
    if (local_rc != 0) return local_rc;  // (after the last collective)
    out->stage = stage;
    out->bad = bad;
    return bad != 0 ? 1 : 0;
}

}  // namespace lh
