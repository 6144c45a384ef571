"""One process per GPU: contiguous sharding of the n-vector and the scalar all-reduce.

Every hot-path op is element-wise or a sum over i, so rank r owns the contiguous index range
``shard_range(n, r, world)`` of x, g, xp, gp, d, pg, wp and all 2m history vectors, and the only
cross-rank traffic is a sum all-reduce of 1-6 f64 scalars per reduction -- RCCL
``ncclAllReduce(ncclDouble, ncclSum)`` enqueued on the compute stream by liblbfgs_hip.so.
`torch.distributed` is used here only to rendezvous (share the RCCL unique id), or -- with
``kind="callback"`` -- as a host all-reduce (gloo) for tests and for machines without RCCL.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _ffi
from .api import Context, LbfgsError

ALIGN = 256  # shard boundaries are multiples of 256 elements (2 KiB): keeps 16-B loads aligned


def shard_range(n, rank, world, align=ALIGN):
    """[lo, hi) of `rank`: ceil(n/world) rounded up to `align`, last ranks may be short or empty."""
    per = -(-n // world)
    per = -(-per // align) * align
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


# ---- the rendezvous: torch.distributed by default.  Anything else that can tell a rank its number and gather / broadcast a small
# Python object works as well -- pass it as `process_group`: an object with `.rank`, `.world`, `.all_gather_object(obj) -> list`
# and `.broadcast_object(obj, src) -> obj` (tools/eight_ranks_one_gpu.py hosts several ranks per process that way).
def _custom(pg):
    return pg is not None and hasattr(pg, "all_gather_object") and hasattr(pg, "world")


def _rank(pg):
    if _custom(pg):
        return pg.rank
    import torch.distributed as dist

    return dist.get_rank(pg)


def _world(pg):
    if _custom(pg):
        return pg.world
    import torch.distributed as dist

    return dist.get_world_size(pg)


def _all_gather(pg, obj):
    if _custom(pg):
        return pg.all_gather_object(obj)
    import torch.distributed as dist

    out = [None] * dist.get_world_size(pg)
    dist.all_gather_object(out, obj, group=pg)
    return out


def _broadcast(pg, obj, src=0):
    if _custom(pg):
        return pg.broadcast_object(obj, src)
    import torch.distributed as dist

    box = [obj]
    dist.broadcast_object_list(box, src=src, group=pg)
    return box[0]


class CommSpec:
    """Keeps the ctypes lbfgs_hip_comm and whatever it points at alive."""

    def __init__(self, kind, unique_id=None, callback=None):
        self.c = _ffi.Comm()
        self.c.kind = kind
        self._id = unique_id
        self._cb = callback
        if unique_id is not None:
            self.c.rccl_unique_id = C.cast(unique_id, C.c_void_p)
        if callback is not None:
            self.c.callback = callback


def rccl_comm(process_group=None, exclusive_device=False):
    """RCCL communicator: rank 0 makes the unique id, torch.distributed broadcasts it.
    exclusive_device=True (one process per GPU, nobody else on it): the two-loop may run as the persistent kernel with the
    all-reduces GATED on a second stream (csrc/stream.h ext_exchange) instead of one kernel per step."""
    _ffi.torch_before_rccl()  # (torch is imported before librccl is touched)
    L = _ffi.load()
    buf = (C.c_char * 128)()
    if _rank(process_group) == 0:
        rc = L.lbfgs_hip_rccl_unique_id(buf)
        if rc != 0:
            raise LbfgsError(rc, L.lbfgs_hip_last_error(None).decode())
    raw = _broadcast(process_group, bytes(buf.raw), 0)
    C.memmove(buf, raw, 128)
    spec = CommSpec(_ffi.COMM_RCCL, unique_id=buf)
    spec.c.exclusive_device = int(bool(exclusive_device))
    return spec


def p2p_comm(device, process_group=None, timeout_s=5.0, exclusive_device=False, placement=_ffi.MAILBOX_AUTO):
    """Direct exchange between the GPUs: every rank exports its mailbox by a 64-byte handle, torch.distributed gathers them.
    exclusive_device=True tells the library that no other rank shares this rank's GPU (lbfgs_hip_comm.exclusive_device):
    only then does it use kernels that fill the whole chip while they wait for their peers.
    placement: MAILBOX_DEVICE (uncached HBM, mapped by the peers through HIP IPC over xGMI), MAILBOX_HOST (a shared-memory
    segment registered with HIP: host-coherent, reached over PCIe) or MAILBOX_AUTO (LBFGS_HIP_P2P_MAILBOX, default device)."""
    L = _ffi.load()
    world = _world(process_group)
    mbox = C.c_void_p()
    hdl = (C.c_char * 64)()
    rc = L.lbfgs_hip_p2p_mailbox_create2(device, placement, C.byref(mbox), hdl)
    ok = _all_gather(process_group, (rc, bytes(hdl.raw)))
    if any(r != 0 for r, _ in ok):
        if rc == 0:
            L.lbfgs_hip_p2p_mailbox_destroy(device, mbox)
        raise LbfgsError(_ffi.HIP_ERR_COMM, "P2P mailbox creation failed on some rank: " +
                         L.lbfgs_hip_last_error(None).decode())
    handles = (C.c_char * (64 * world))()
    for r, (_, h) in enumerate(ok):
        C.memmove(C.addressof(handles) + 64 * r, h, 64)
    spec = CommSpec(_ffi.COMM_P2P)
    spec._handles = handles
    spec.c.p2p_mailbox = mbox
    spec.c.p2p_handles = C.cast(handles, C.c_void_p)
    spec.c.p2p_timeout_s = timeout_s
    spec.c.exclusive_device = int(bool(exclusive_device))
    return spec


def _p2p_context(n, device, shard, process_group, stream, exclusive_device, placement):
    """The P2P context of this rank, created COLLECTIVELY: every rank learns whether every rank succeeded (one rank that
    cannot map a peer's mailbox must not leave the others with a context nobody will answer in).  Returns
    (ctx or None, this rank's error or None)."""
    ctx, err = None, None
    try:
        comm = p2p_comm(device, process_group, exclusive_device=exclusive_device, placement=placement)
        ctx = Context(n, device=device, shard=shard, comm=comm, stream=stream)
    except LbfgsError as e:
        err = e
    oks = _all_gather(process_group, err is None)
    if all(oks):
        # every rank has mapped every mailbox: a host-placed segment can lose its name now (nothing outlives the processes)
        placed = C.c_int(-1)
        ctx.check(ctx._L.lbfgs_hip_ctx_p2p_seal(ctx._h, C.byref(placed)))
        ctx.p2p_placement = {0: "device", 1: "host"}.get(placed.value)
        return ctx, None
    if ctx is not None:
        ctx.close()
    return None, err or LbfgsError(_ffi.HIP_ERR_COMM, "the P2P communicator failed on another rank")


def callback_comm(process_group=None):
    """Host all-reduce through torch.distributed (gloo or nccl group): slow, for tests."""
    if _custom(process_group):  # (sums in rank order: the same bits on every rank)
        def allreduce(_user, buf, count):
            try:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                parts = process_group.all_gather_object(a.copy())
                tot = parts[0].copy()
                for p in parts[1:]:
                    tot += p
                a[:] = tot
                return 0
            except Exception:
                return 1

        return CommSpec(_ffi.COMM_CALLBACK, callback=_ffi.ALLREDUCE_CB(allreduce))

    import torch
    import torch.distributed as dist

    def allreduce(_user, buf, count):
        try:
            a = np.ctypeslib.as_array(buf, shape=(count,))
            t = torch.from_numpy(a.copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
            a[:] = t.numpy()
            return 0
        except Exception:
            return 1

    return CommSpec(_ffi.COMM_CALLBACK, callback=_ffi.ALLREDUCE_CB(allreduce))


def sharded_context(n, device=0, kind="rccl", process_group=None, stream=None, exclusive_device=False, bounds=None):
    """Context for this rank's shard of a global n-vector (world from torch.distributed).  exclusive_device: see p2p_comm.
    bounds: explicit shard boundaries [0, b_1, ..., b_{world-1}, n] (non-decreasing; default: shard_range's even split)."""
    rank, world = _rank(process_group), _world(process_group)
    if bounds is not None:
        if len(bounds) != world + 1 or bounds[0] != 0 or bounds[-1] != n or any(a > b for a, b in zip(bounds, bounds[1:])):
            raise ValueError("bounds must be [0, ..., n], non-decreasing, one entry more than there are ranks")
        lo, hi = bounds[rank], bounds[rank + 1]
    else:
        lo, hi = shard_range(n, rank, world)
    shard = _ffi.Shard(rank, world, n, lo, hi - lo)
    if world == 1 and not (kind == "rccl" and os.environ.get("LBFGS_FORCE_RCCL") == "1"):
        comm = None  # (LBFGS_FORCE_RCCL=1 exercises the RCCL code path with a 1-rank communicator)
    elif kind == "rccl":
        comm = rccl_comm(process_group, exclusive_device=exclusive_device)
    elif kind == "callback":
        comm = callback_comm(process_group)
    elif kind in ("p2p", "p2p-device", "p2p-host"):
        # "p2p": mailboxes in device memory, mapped by the peers through HIP IPC (xGMI) -- unless LBFGS_HIP_P2P_MAILBOX says
        # otherwise -- and, if ANY rank fails there (hipIpcOpenMemHandle of a peer's device memory refused), every rank
        # retries with host-placed mailboxes (shared memory, PCIe): same kernels, same bits, more latency per exchange.
        # "p2p-device" / "p2p-host" force one placement, no retry.
        first = {"p2p": _ffi.MAILBOX_AUTO, "p2p-device": _ffi.MAILBOX_DEVICE, "p2p-host": _ffi.MAILBOX_HOST}[kind]
        ctx, err = _p2p_context(n, device, shard, process_group, stream, exclusive_device, first)
        if ctx is None and kind == "p2p" and os.environ.get("LBFGS_HIP_P2P_MAILBOX", "device") != "host":
            if rank == 0:
                import sys

                print(f"[lbfgs_hip] P2P mailboxes in device memory are not usable here ({err}); retrying with host-placed "
                      "mailboxes", file=sys.stderr)
            ctx, err = _p2p_context(n, device, shard, process_group, stream, exclusive_device, _ffi.MAILBOX_HOST)
        if ctx is None:
            raise err
        return ctx
    else:
        raise ValueError(kind)
    return Context(n, device=device, shard=shard, comm=comm, stream=stream)
