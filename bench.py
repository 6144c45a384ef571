#!/usr/bin/env python3
"""bench.py -- L-BFGS iterations/sec and two-loop HBM GB/s on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dim 100000000] [--hist 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE L-BFGS iteration (LbfgsState::propagate, reference src/lbfgs.rs:503-560) of the
synthetic diagonal quadratic of BASELINE.json config 4: n = 1e8, m = 10, More-Thuente, f64, the
crate's default parameters, everything resident in HBM (device objective, no PCIe in the timed
region).  One step = the line search (per trial one pass over xp and d that returns f and g.d), the
history update (which also forms the accepted x and g) and the fused two-loop recursion.  With N > 1
the n-vector is sharded contiguously over the ranks (total work fixed => "strong" scaling) and every
reduction is closed by an all-reduce of its f64 scalars.

Process structure.  N = 1 runs in this process (its CPU baseline runs BESIDE the GPU work, in a child pinned to one
core).  Its line EXISTS from the moment the timed measurement returns; everything that follows -- the vector-free extension,
three rocprofv3 child runs (each <= --rocprof-timeout, 60 s; skipped when less than that is left), the wait for the baseline
child -- runs inside ONE budget (--post-budget, 300 s, and never past --total-budget from the start of the process), and
SIGTERM / SIGINT end the running child and print the line measured so far; `config.budget.cut`, `cpu_baseline.sample` and
`roofline.traffic_live_error` / `rocprofv3_error` say what a line lacks and why.  N > 1 never measures in the process the user (or torch.distributed.run) started: that process is a SUPERVISOR
that touches no GPU and runs fresh child jobs, each under a wall-clock timeout and all of them inside ONE total budget
(--total-budget, default 520 s: the driver allows 600):
  * `python bench.py --gpus N` (no RANK in the environment): per job one
    `python -m torch.distributed.run --nproc-per-node N bench.py --_rank-mode --comm <leg>` child;
  * launched by torch.distributed.run (RANK set): every rank is a supervisor (gloo group, CPU only);
    rank 0 decides, picks a fresh rendezvous port per job, and each rank starts ITS child
    `python bench.py --_rank-mode --comm <leg>` with the same RANK / LOCAL_RANK / WORLD_SIZE.
Phase 1, PROBES: per communicator one short job (<= --probe-timeout) that only creates the context and runs its
start-up self-test (known-answer reductions, a reduction closed inside a streaming kernel, a whole two-loop recursion with
an exact answer).  Phase 2, MEASUREMENTS, only for communicators whose probe passed, each with a share of what is left of
the budget: "p2p" (direct exchange inside the reducing kernels, mailboxes in device memory mapped over xGMI; the two-loop
runs as one persistent kernel), "p2p-per-step" (the same communicator with a kernel per two-loop step) only if the p2p
MEASUREMENT failed, "rccl" (ncclAllReduce; inside the two-loop gated on a second stream under the persistent kernel, and
"rccl-per-step" -- a kernel per step, every all-reduce on the compute stream -- only if that measurement failed), "p2p-host" (the p2p exchange with host-coherent
mailboxes: the placement a machine falls back to when device memory cannot be mapped between its GPUs; tried before rccl
when p2p gave nothing), "callback" (host-staged all-reduce through gloo) only if nothing produced a result.  Every job's
outcome is echoed to stderr as it lands; SIGTERM / SIGINT kill the running job and print the best line so far.  The run
prints ONE JSON line with the best leg as `value`, every probe and leg in `config.probes` / `config.legs`, and exits 0 if
any leg succeeded.  The line SAYS which leg that is: `metric` ends in "scalars closed by: <leg in words>", `config.allreduce`
/ `config.allreduce_says` name it, and `config.rccl` carries the RCCL leg's iterations/sec, two-loop ms and ncclAllReduce
microseconds at the top level of `config` whichever leg won (the north star's wording is "scalar RCCL all-reduce").

Timing.  W warm-up steps (plus whatever fills the history: bound = m before anything is timed), then
EXACTLY K steps between barrier + synchronize on both sides, max over ranks.  That K-step region is
repeated --repeats times, each time from a freshly built state brought to the same iteration, so
every repeat times the same K iterations; `value` / `ms_per_step` are the MEDIAN repeat, all repeats
are listed in `config.repeats_iters_per_sec`.

The JSON line carries, besides the contract fields:
  roofline      the dominant kernel, timed with HIP events on the launch stream over the timed region, against the 8 TB/s
                HBM peak.  By default that is the WHOLE two-loop recursion as one persistent kernel
                (two_loop_resident_kernel, csrc/resident.h): the running vector q stays in registers + LDS, so the on-chip
                elements cost (4m+1) n-vector passes (g once, every s and y twice, d written once) and -- hybrid form, shards
                beyond ~1.25e7 elements -- the rest of q, kept in HBM, the kernel-per-step (8m-1); `bytes_per_launch` is that
                sum.  Only with LBFGS_HIP_RESIDENT=0 is it the two-loop step kernel (q += c*u ; out = v.q: 3 reads + 1 write =
                32 bytes per element).  `note` bounds from above what the Infinity Cache can absorb of those bytes;
  cpu_baseline  the CPU oracle (reference operation order, 1 thread) on the metric's own
                configuration (n = 1e8, m = 10: 3 iterations after m+2 warm-up, ~22 GB of host memory) when
                the host has the memory, with the sampled-and-scaled figure beside it.  N = 1: a child of the
                measuring process; N > 1: a child of rank 0's SUPERVISOR (which touches no GPU), timed beside the legs.
and, so that an N > 1 line can be cross-checked and a scaling shortfall attributed (lbfgs_hip_ctx_comm_info):
  config.comm_info / config.ranks_seen   what the communicator really spans: RCCL's own ncclCommCount (context creation
                fails unless it equals the shard's world), or the P2P mailboxes rank 0 mapped and their placement;
  roofline.exchange_us_mean, roofline.exchanges_per_two_loop, config.legs[leg].*   one cross-rank exchange inside a
                two-loop as the device timed it (P2P: the exchanging workgroup's wall clock; RCCL: HIP events around
                the all-reduce launches) and how many a two-loop made, over the timed regions only;
  roofline.traffic   N = 1: HBM bytes per launch of the dominant kernel taken BY THIS RUN -- after the timed region two child
                runs of this command under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace
                only; a second or two each; --no-live-traffic skips them), `traffic_source` says so and `traffic_committed` keeps the
                figure of the committed passes (profiles/pmc_traffic*.json) beside it.  Without the live passes (N > 1, under
                a profiler, rocprofv3 missing or failing: `traffic_live_error`) the committed figure is quoted, with
  roofline.traffic_build_id / traffic_is_current   which build the counter passes behind `traffic` were made
                with, and whether it is the build measuring now.
  roofline.rocprofv3_avg_ms   N = 1: the dominant kernel's duration as a `rocprofv3 --kernel-trace --stats` child run of this
                command sees it (full-depth dispatches), beside the HIP-event `avg_ms` (`rocprofv3_over_hip_events`).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
METRIC = "L-BFGS iters/sec (two-loop HBM GB/s in roofline) at n=1e8, m=10"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--dim", dest="n", type=int, default=100_000_000, help="n, the number of variables")
    ap.add_argument("--hist", dest="m", type=int, default=10, help="m, the number of L-BFGS corrections")
    ap.add_argument("--repeats", type=int, default=0,
                    help="how many times the K-step timed region is measured (median reported); 0 = as many as it takes to "
                         "time --min-timed-seconds of iterations, at least 5 and at most 40")
    ap.add_argument("--min-timed-seconds", type=float, default=5.0,
                    help="--repeats 0: total length of the timed regions (the GPU is busy about twice as long: every repeat "
                         "rebuilds its state and warms up)")
    ap.add_argument("--no-prof", action="store_true", help="do not time kernels with HIP events in the timed region")
    ap.add_argument("--prof-every", type=int, default=5,
                    help="time kernels with HIP events on every k-th step of the timed region only: an event pair per "
                         "kernel costs ~1.5 us of stream time, 11 %% of an iteration at 100 MB shards when always on -- and "
                         "pairs around EVERY kernel read the long kernels ~6 %% short (the time reappears between the "
                         "pairs; rocprofv3 agrees with the sparse sampling, profiles/EXPERIMENTS.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=20_000_000, help="sample size of the sampled CPU baseline (~10 s on one core)")
    ap.add_argument("--no-cpu-full", action="store_true",
                    help="skip the CPU baseline at the metric's own n (needs (2m+7)*8n bytes of host memory, ~50 s)")
    ap.add_argument("--grid", type=int, default=0, help="workgroups per launch (0 = library default)")
    ap.add_argument("--comm", default="auto", choices=["auto", "rccl", "p2p", "p2p-host", "callback"],
                    help="N>1: how scalars are all-reduced (auto = p2p, rccl and p2p-host, each probed first; callback if none works)")
    ap.add_argument("--pg-backend", default="auto",
                    help="torch.distributed backend of a rank process, used for rendezvous, barriers and the max over "
                         "ranks (auto = nccl for the rccl leg, gloo otherwise)")
    ap.add_argument("--leg-timeout", type=float, default=150.0,
                    help="N>1: upper bound on one measurement job (s); the actual bound is its share of what is left of --total-budget")
    ap.add_argument("--total-budget", type=float, default=520.0,
                    help="wall-clock budget of the whole run (s; the driver allows 600).  N>1: probes and legs get their timeouts from "
                         "what is left of it.  N=1: nothing that FOLLOWS the measurement (counter passes, waiting for the CPU baseline) "
                         "may run past it")
    ap.add_argument("--post-budget", type=float, default=300.0,
                    help="N=1: wall-clock budget (s) of everything that follows the timed measurement -- the vector-free extension, the "
                         "rocprofv3 child runs, the wait for the CPU baseline child: ONE budget; a pass is skipped when less than its "
                         "timeout is left, and the line is printed when it is spent")
    ap.add_argument("--rocprof-timeout", type=float, default=60.0,
                    help="N=1: bound on ONE rocprofv3 child run (s; they take a few seconds)")
    ap.add_argument("--probe-timeout", type=float, default=30.0,
                    help="N>1: bound on one communicator probe (s); the first probe and rccl's get twice that (the box's cold start; RCCL's first communicator)")
    ap.add_argument("--no-vector-free", action="store_true",
                    help="skip the extra measurement of the vector-free (Gram) two-loop extension")
    ap.add_argument("--line-eval", type=int, default=2, choices=[0, 1, 2],
                    help="lbfgs_evaluator.fuse_line_eval: 2 = trials write no vectors (default), 1 = every trial writes x and g, "
                         "0 = separate passes")
    ap.add_argument("--device", type=int, default=-1, help="force a device index (testing: several ranks on one GPU)")
    ap.add_argument("--exclusive-device", type=int, default=-1, choices=[-1, 0, 1],
                    help="tell the communicator that every rank owns its GPU (-1: yes unless --device is given); experiments "
                         "with several ranks on ONE GPU pass 1 together with LBFGS_HIP_RESIDENT_GRID = CUs / ranks")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not re-take roofline.traffic with two rocprofv3 --pmc child runs of this command after the "
                         "measurement (the committed counter passes of profiles/ are quoted instead)")
    ap.add_argument("--_rank-mode", dest="rank_mode", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--_probe", dest="probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--_cpu-baseline-child", dest="cpu_child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ======================================================================================== CPU baseline
def host_info():
    model, cores = "unknown", os.cpu_count() or 0
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                avail = int(ln.split()[1]) * 1024
                break
    except OSError:
        pass
    usable = cores
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    return model, cores, usable, avail


def pin_to_one_core():
    """Pin this process to ONE of the cores it may use (the last one: furthest from core 0, where the kernel and the GPU
    driver's interrupt handling tend to live).  -> the core, or None where the platform has no affinity call."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, {cores[-1]})
        return cores[-1]
    except (AttributeError, OSError, IndexError):
        return None


def oracle_ips(n, m, timed):
    """The oracle (C restatement of the reference's sequential arithmetic) on one host core: `timed` iterations, each timed
    on its own, after the history has filled.  -> (iterations/sec of the MEDIAN iteration, of the FASTEST one, warm-up count)."""
    import numpy as np

    from oracle import oracle as O

    x = np.zeros(n)
    st = O.lbfgs().with_m(m).with_epsilon(0.0).build(x, O.quadratic())
    warm = m + 2  # the history is full (bound = m) from iteration m+2 on
    for _ in range(warm):
        st.propagate()
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        st.propagate()
        ts.append(time.perf_counter() - t0)
    st.close()
    ts.sort()
    med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    return 1.0 / med, 1.0 / ts[0], warm


def cpu_baseline(a):
    model, cores, usable, avail = host_info()
    core = pin_to_one_core()
    ips_s, best_s, warm_s = oracle_ips(a.cpu_n, a.m, 6)
    scaled = ips_s * a.cpu_n / a.n
    out = {
        "value": scaled, "unit": "iters/sec", "cores": 1, "kind": "port",
        "host_cpu": model, "host_cores_total": cores, "host_cores_usable": usable, "pinned_to_core": core,
        "value_is": "1 / (median of the separately timed iterations)",
        "sampled_and_scaled": {"value": scaled, "best": best_s * a.cpu_n / a.n, "n_sample": a.cpu_n,
                               "iters_per_sec_at_n_sample": ips_s, "timed_iterations": 6, "warmup_iterations": warm_s},
        "sample": f"oracle (gcc -O2 -ffp-contract=off, sequential sums, 1 thread pinned to one core) on the same quadratic at "
                  f"n={a.cpu_n}, m={a.m}: median of 6 iterations after {warm_s} warm-up = {ips_s:.3f} iters/sec, scaled by "
                  f"n_sample/n (every pass is O(n)) to n={a.n}",
    }
    need = (2 * a.m + 9) * 8 * a.n  # 2m history + 7 problem vectors (+ the caller's x and slack)
    if a.no_cpu_full or a.n <= a.cpu_n:
        return out
    if avail is None or avail < need * 1.25:
        out["full_size"] = f"skipped: needs {need / 1e9:.1f} GB of host memory, {0 if avail is None else avail / 1e9:.1f} GB available"
        return out
    try:
        ips_f, best_f, warm_f = oracle_ips(a.n, a.m, 3)
    except Exception as e:  # noqa: BLE001  (MemoryError included)
        out["full_size"] = f"failed: {e!r}"
        return out
    out["value"] = ips_f
    out["full_size"] = {"value": ips_f, "best": best_f, "n": a.n, "timed_iterations": 3, "warmup_iterations": warm_f}
    out["sample"] = (f"oracle (gcc -O2 -ffp-contract=off, sequential sums, 1 thread pinned to one core) on the metric's own "
                     f"configuration n={a.n}, m={a.m}: median of 3 separately timed iterations after {warm_f} warm-up = "
                     f"{ips_f:.4f} iters/sec (fastest: {best_f:.4f}; sampled at n={a.cpu_n} and scaled: {scaled:.4f}); timed in a "
                     f"child process while the GPU measurement ran")
    return out


def cpu_baseline_child_main(a):
    """`bench.py --_cpu-baseline-child`: the CPU baseline alone, as ONE JSON line (the N = 1 run starts this before its GPU
    work and collects the line afterwards, so the ~45 s of CPU time overlap the GPU measurement instead of following it)."""
    out = cpu_baseline(a)
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
    return 0


def start_cpu_baseline(a):
    """-> Popen of the child above (never touches a GPU), or None."""
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--_cpu-baseline-child", "--dim", str(a.n), "--hist", str(a.m),
            "--cpu-n", str(min(a.cpu_n, a.n))]
    if a.no_cpu_full:
        args.append("--no-cpu-full")
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    if os.environ.get("LBFGS_BENCH_TEST_CPU_CHILD") == "hang":  # test hook: a baseline child that never returns
        args = [sys.executable, "-c", "import time; time.sleep(100000)"]
    try:
        return subprocess.Popen(args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    except OSError as e:
        print(f"[bench] cpu baseline child not started: {e}", file=sys.stderr)
        return None


def cpu_baseline_missing(why):
    return {"value": None, "unit": "iters/sec", "cores": 1, "kind": "port", "sample": why}


def collect_cpu_baseline(child, a, timeout=420.0):
    if child is None:
        return cpu_baseline(a)  # in this process, after the GPU work
    try:
        out, _ = child.communicate(timeout=timeout)
        j = last_json(out)
        if j is not None and "value" in j:
            return j
        return cpu_baseline_missing(f"the baseline child printed no result (exit code {child.returncode})")
    except subprocess.TimeoutExpired:
        child.kill()
        try:
            child.communicate(timeout=5)
        except Exception:  # noqa: BLE001
            pass
        return cpu_baseline_missing(f"cut: the baseline child (the oracle on one pinned core, ~60 s at the metric's size) had not finished "
                                    f"when the run's budget ended (waited {timeout:.0f} s more after the GPU work)")


# ======================================================================================== one rank
def pick_device(local_rank, forced, visible):
    """The device index of this rank.  One process per GPU: LOCAL_RANK -- unless the launcher has masked the devices per rank
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = one GPU each), when every rank sees fewer devices than there are local ranks
    and its own is among the ones it sees: LOCAL_RANK modulo that count (0 with one device each).  Such ranks cannot map each
    other's device memory: the "p2p" probe then fails with the IPC error and the run goes on with host-placed mailboxes and RCCL.
    `forced` (--device, tests: several ranks on one card) wins."""
    if forced >= 0:
        return forced
    if visible > 0 and local_rank >= visible:
        return local_rank % visible
    return local_rank


class Env:
    """rank / world / torch.distributed plumbing (only rendezvous, barriers and a max over ranks)."""

    def __init__(self, a, comm):
        self.a = a
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dev = self.local_rank if a.device < 0 else a.device
        self.dist = self.torch = None
        self.backend = a.pg_backend
        if self.backend == "auto":
            self.backend = "nccl" if comm == "rccl" else "gloo"
        if self.world > 1 or "RANK" in os.environ:  # one rank per GPU
            import torch
            import torch.distributed as dist

            self.torch, self.dist = torch, dist
            try:  # (counting devices does not initialise the GPU)
                visible = int(torch.cuda.device_count())
            except Exception:  # noqa: BLE001
                visible = 0
            dev = pick_device(self.local_rank, a.device, visible)
            if dev != self.dev:
                print(f"[bench] rank {self.rank}: LOCAL_RANK {self.local_rank} but {visible} visible device(s): the launcher masks "
                      f"devices per rank; taking device {dev}", file=sys.stderr)
                self.dev = dev
            torch.cuda.set_device(self.dev)
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.dev))
            else:
                dist.init_process_group(self.backend)

    def _tensor(self, v):
        return self.torch.tensor([v], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")

    def barrier(self, ctx=None):
        if ctx is not None:
            try:
                ctx.sync()
            except Exception:  # noqa: BLE001
                pass
        if self.dist is not None:
            self.torch.cuda.synchronize()
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def reduce(self, v, op):
        if self.dist is None:
            return v
        t = self._tensor(v)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def make_context(env, kind):
    """-> (ctx or None, label).  Every rank returns the same decision."""
    import rust_lbfgs_amd as R
    from rust_lbfgs_amd.dist import sharded_context

    a = env.a
    if env.world == 1:
        if env.dist is not None and os.environ.get("LBFGS_FORCE_RCCL") == "1":  # (the RCCL code path with a 1-rank communicator)
            return sharded_context(a.n, device=env.dev, kind="rccl",
                                   exclusive_device=(a.device < 0) if a.exclusive_device < 0 else bool(a.exclusive_device)), "rccl"
        return R.Context(a.n, device=env.dev), "none"
    ok, ctx = 1.0, None
    try:
        # one rank per GPU (the driver's launch) unless --device forces several ranks onto one card (tests).
        # The "p2p" leg means mailboxes in DEVICE memory, strictly (no silent retry: host placement is its own leg).
        ctx = sharded_context(a.n, device=env.dev, kind={"p2p": "p2p-device"}.get(kind, kind),
                              process_group=getattr(env, "group", None),  # (None: torch.distributed; tools/eight_ranks_one_gpu.py brings its own)
                              exclusive_device=(a.device < 0) if a.exclusive_device < 0 else bool(a.exclusive_device))
        # known-answer reductions through the real code path before trusting it
        tri = env.world * (env.world + 1) / 2.0
        for it in range(16):
            ctx.set_scalars(200, [float((env.rank + 1) * (it + 1)), float(env.rank == it % env.world),
                                  -0.5 * (env.rank + 1)])
            ctx.check(ctx._L.lbfgs_hip_scalars_allreduce(ctx._h, 200, 3))
            if list(ctx.scalars(200, 3)) != [tri * (it + 1), 1.0, -0.5 * tri]:
                ok = 0.0
        # ... and a reduction closed INSIDE a streaming kernel (p2p: by its last workgroup; rccl: on-stream after it):
        # sum over ranks of n_local * (rank+1)^2, exact in f64 in any summation order
        from rust_lbfgs_amd.dist import shard_range
        from rust_lbfgs_amd.math import DeviceVec

        u = DeviceVec(ctx)
        try:
            want = 0.0
            for r in range(env.world):
                lo, hi = shard_range(a.n, r, env.world)
                want += float(hi - lo) * (r + 1) ** 2
            for it in range(4):
                u.fill(float(env.rank + 1))
                if u.vecdot(u) != want:
                    ok = 0.0
        finally:
            u.free()
        # ... and a whole two-loop recursion with an EXACT answer, through whatever launch form this context takes (ranks
        # that own their GPU under p2p: the persistent kernel, whose hand-offs carry the exchange -- a path the reductions
        # above do not touch).  m = 1, s = (1,0,1,0,..), y = (2,2,..), g = (-1,-3,-1,-3,..): alpha = 1/2, gamma = 1/4,
        # beta = 1/2, d = (0, 1/2, 0, 1/2, ..), ||d||^2 = n/8, g.d = -3n/4 -- every sum is a sum of dyadic numbers.
        if a.n % 2 == 0:
            import numpy as np

            from rust_lbfgs_amd import hotpath as H

            nl = ctx.n_local
            hist = H.History(ctx, 1)
            gv, dv = DeviceVec(ctx), DeviceVec(ctx)
            try:
                pat = np.zeros(nl)
                pat[0::2] = 1.0
                hist.s(0).upload(pat)
                hist.y(0).fill(2.0)
                pat[0::2], pat[1::2] = -1.0, -3.0
                gv.upload(pat)
                hist.set_scalars(ys=np.array([float(a.n)]), alpha=np.zeros(1))
                ctx.set_scalars(7, [float(a.n), 4.0 * a.n])          # gamma = ys / yy
                for _ in range(2):
                    hist.two_loop(dv, gv, 1, 0, 7, 8, 12)
                    dn2, gd = ctx.scalars(12, 2)
                    dh = dv.to_numpy()
                    if dn2 != a.n / 8.0 or gd != -0.75 * a.n or np.any(dh[0::2] != 0.0) or np.any(dh[1::2] != 0.5):
                        print(f"[bench] rank {env.rank}: two-loop self-test: ||d||^2 = {dn2} (want {a.n / 8.0}), "
                              f"g.d = {gd} (want {-0.75 * a.n})", file=sys.stderr)
                        ok = 0.0
            finally:
                hist.free()
                gv.free()
                dv.free()
    except Exception as e:  # noqa: BLE001
        print(f"[bench] rank {env.rank}: {kind} communicator unavailable: {e}", file=sys.stderr)
        ok = 0.0
    if env.reduce(ok, "MIN") != 1.0:
        if ctx is not None:
            ctx.close()
        return None, kind
    return ctx, kind


def loaded_build_id():
    """lbfgs_hip_build_id() of the library this process measures with"""
    from rust_lbfgs_amd import _ffi

    try:
        return _ffi.load().lbfgs_hip_build_id().decode()
    except Exception:  # noqa: BLE001
        return None


def traffic_lookup(n_local, m, kernel, build_id=None, profiles_dir=None, resident_elements=None):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS
    command (tools/profile_round.sh; rocprofv3 counters cannot be collected from inside this process).  A file counts only
    if it was taken from `bench.py` itself (its `command` stamp: tools/profile_configs.sh profiles tools/run_configs.py --
    other objectives, OWL-QN, damping -- at sizes this program can also be asked to run), at this shard size, this m and
    for this kernel (strict: a near-by size is a different measurement); the record names the build the passes were made
    with and whether that is the build measuring now."""
    import glob

    if build_id is None:
        build_id = loaded_build_id()
    out, best = {}, None
    for path in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "pmc_traffic*.json"))):
        try:
            pm = json.load(open(path))
            if (pm.get("command") == "bench.py" and pm["n_local"] == n_local and pm.get("m", 10) == m
                    and pm.get("kernel", "stream_kernel").split("<")[0] == kernel.split("<")[0]):
                # (the persistent kernel on another grid keeps another share of q on the chip and moves other bytes: rehearsals
                # of several ranks on one GPU must not be given the whole-GPU figure)
                if resident_elements is not None and pm.get("resident_elements") not in (None, resident_elements):
                    continue
                cur = pm.get("build_id") is not None and pm.get("build_id") == build_id
                if best is None or (cur and not best[0]):
                    best = (cur, pm, path)
        except Exception:  # noqa: BLE001
            pass
    if best is not None:
        cur, pm, path = best
        out = {"traffic": pm["traffic_bytes_per_launch"] / 1e9,
               "traffic_unit": "GB per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes)",
               "traffic_source": pm.get("_source"), "traffic_file": os.path.relpath(path, ROOT),
               "traffic_build_id": pm.get("build_id"), "traffic_is_current": bool(cur), "loaded_build_id": build_id}
    return out


def live_traffic_wanted(a):
    """N = 1 only, not inside a profiler (tools/profile_round.sh runs this program under rocprofv3), not in a child of itself."""
    if a.no_live_traffic or a.no_prof or os.environ.get("LBFGS_BENCH_LIVE_TRAFFIC", "1") == "0":
        return False
    if os.environ.get("LBFGS_BENCH_LIVE_TRAFFIC") == "force":  # (the CPU suite: the orchestration on the test double)
        return True
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or "RANK" in os.environ or a.gpus > 1:
        return False
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock" or os.environ.get("LBFGS_BENCH_WORKER"):  # (the CPU test double)
        return False
    return not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def rocprof_child(a, rocprof_args, outdir, steps, timeout_s):
    """One run of this very command (same size, `steps` timed steps after the warm-up, nothing else) under rocprofv3 with
    `rocprof_args`, output under `outdir`.  -> None, or what went wrong."""
    import shutil

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return "rocprofv3 not found"
    env = dict(os.environ, TMPDIR="/tmp", LBFGS_BENCH_LIVE_TRAFFIC="0")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--dim", str(a.n), "--hist", str(a.m), "--steps", str(steps), "--warmup",
             str(max(a.warmup, a.m + 2)), "--repeats", "1", "--line-eval", str(a.line_eval), "--no-cpu-baseline", "--no-prof",
             "--no-vector-free", "--no-live-traffic"]
    if a.grid:
        child += ["--grid", str(a.grid)]
    cmd = [exe] + list(rocprof_args) + ["--output-format", "csv", "-d", outdir, "--"] + child
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
    Job.proc = p  # (a signal handler ends it with its whole group)
    try:
        _, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)  # (the group this call started: rocprofv3 and the program under it)
        except OSError:
            pass
        p.communicate()
        return f"did not finish within {timeout_s:.0f} s"
    finally:
        Job.proc = None
    if p.returncode != 0:
        return f"exited with {p.returncode}: {err[-300:]}"
    return None


def _summariser():
    import importlib.util

    spec = importlib.util.spec_from_file_location("summarize_profile", os.path.join(ROOT, "tools", "summarize_profile.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    return sp


def budget_says(left, timeout_s):
    """-> None if a child run bounded by `timeout_s` still fits into what is left of the run's budget, else the sentence the record carries"""
    if left is None or left() >= timeout_s:
        return None
    return (f"skipped: {max(left(), 0.0):.0f} s of the run's budget left, a pass is bounded by {timeout_s:.0f} s "
            f"(--post-budget / --total-budget / --rocprof-timeout)")


def live_kernel_time(a, roof, steps=40, timeout_s=60.0, left=None):
    """The dominant kernel's duration as rocprofv3 sees it (`--kernel-trace --stats`, a child run of this command: begin to end of
    the dispatch), beside the HIP-event figure of the timed region (`roofline.avg_ms`: launch to completion on the stream, so it
    also holds the dispatch gap in front of the kernel) -- the two must agree to a few per cent, and the record shows both."""
    import shutil
    import tempfile

    cut = budget_says(left, timeout_s)
    if cut:
        return {"rocprofv3_avg_ms": None, "rocprofv3_error": "the --kernel-trace --stats pass " + cut}
    work = tempfile.mkdtemp(prefix="lbfgs_bench_pmc_", dir="/tmp")
    try:
        print("[bench] roofline.rocprofv3_avg_ms: rocprofv3 --kernel-trace --stats pass of this command (a child run of a few seconds)",
              file=sys.stderr)
        err = rocprof_child(a, ["--kernel-trace", "--stats"], os.path.join(work, "stats"), steps, timeout_s)
        if err:
            return {"rocprofv3_avg_ms": None, "rocprofv3_error": "the --kernel-trace --stats pass " + err}
        durs = _summariser().trace_durations(os.path.join(work, "stats"))
        want = "two_loop_resident_kernel<" if "resident" in roof["kernel"] else "OpTwoLoopStep<false, false, 0"
        keys = [k for k in durs if want in k and durs[k]]
        if not keys:
            return {"rocprofv3_avg_ms": None, "rocprofv3_error": f"no dispatches of {roof['kernel']} in the kernel trace"}
        v = durs[keys[0]]
        ref = sorted(v)[len(v) // 2]  # (most launches are full-depth: the median is one; see tools/summarize_profile.py)
        full = [x for x in v if abs(x - ref) <= 0.05 * ref]
        avg_ms = sum(full) / len(full) / 1e3
        out = {"rocprofv3_avg_ms": avg_ms, "rocprofv3_launches": len(full), "rocprofv3_kernel": keys[0],
               "rocprofv3_says": "average duration of the kernel's full-depth dispatches in a `rocprofv3 --kernel-trace --stats` child "
                                 f"run of this command ({steps} steps after the warm-up)"}
        if roof.get("avg_ms") and roof.get("bytes_per_launch"):
            out["rocprofv3_over_hip_events"] = avg_ms / roof["avg_ms"]
            out["frac_on_rocprofv3_time"] = roof["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def live_traffic(a, roof, timeout_s=60.0, left=None):
    """roofline.traffic taken BY this run: two child runs of this very command under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes, --kernel-trace only: /opt/skills/guides/MI355X_MICROARCH.md, HBM section), a few
    iterations each with the history full, after the timed region and after this process has given its GPU memory back.
    FETCH_SIZE x2 (gfx950), KiB; averaged over the full-depth launches of the dominant kernel (tools/summarize_profile.py:
    the same arithmetic as the committed passes).  Counters cannot be read from inside the measuring process, and a profiled
    run is not a timed run -- hence children.  Any failure leaves the committed look-up in place and says why."""
    import shutil
    import tempfile

    sp = _summariser()
    work = tempfile.mkdtemp(prefix="lbfgs_bench_pmc_", dir="/tmp")
    got, t0 = {}, time.monotonic()
    try:
        for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
            cut = budget_says(left, timeout_s)
            if cut:
                return {"traffic_live": None, "traffic_live_error": f"the {counter} pass " + cut}
            print(f"[bench] roofline.traffic: rocprofv3 --pmc {counter} pass of this command (a child run of a few seconds)", file=sys.stderr)
            err = rocprof_child(a, ["--pmc", counter, "--kernel-trace"], os.path.join(work, sub), 4, timeout_s)
            if err:
                return {"traffic_live": None, "traffic_live_error": f"the {counter} pass " + err}
            per_kernel = sp.pmc(os.path.join(work, sub), counter)
            want = "two_loop_resident_kernel<" if "resident" in roof["kernel"] else "OpTwoLoopStep<false, false, 0"
            keys = [k for k in per_kernel if want in k]
            if not keys:
                return {"traffic_live": None, "traffic_live_error": f"no {counter} rows for {roof['kernel']}"}
            got[counter] = per_kernel[keys[0]]
            got["kernel"] = keys[0]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    rd, wr = got["FETCH_SIZE"] * 1024 * 2, got["WRITE_SIZE"] * 1024
    return {"traffic_live": (rd + wr) / 1e9, "traffic_live_read_GB": rd / 1e9, "traffic_live_write_GB": wr / 1e9,
            "traffic_live_kernel": got["kernel"], "traffic_live_seconds": round(time.monotonic() - t0, 1),
            "traffic_live_source": "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (--kernel-trace only) in two separate child "
                                   "runs of `python3 bench.py` at this size (4 timed steps after the warm-up, history full), "
                                   "averaged over the full-depth dispatches of the kernel; FETCH_SIZE x2 (gfx950 correction of "
                                   "MI355X_MICROARCH.md section HBM), KiB"}


def apply_live_traffic(roof, live):
    """The live figure becomes roofline.traffic; what the committed passes say stays beside it (traffic_committed*)."""
    roof.update({k: v for k, v in live.items()})
    if live.get("traffic_live") is None:
        return roof
    if roof.get("traffic") is not None:
        roof["traffic_committed"] = roof["traffic"]
        roof["traffic_committed_file"] = roof.get("traffic_file")
        roof["traffic_committed_build_id"] = roof.get("traffic_build_id")
    build = roof.get("loaded_build_id") or loaded_build_id()
    roof.update(traffic=live["traffic_live"], traffic_source=live["traffic_live_source"], traffic_file=None,
                traffic_unit="GB per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes)",
                traffic_build_id=build, traffic_is_current=True, loaded_build_id=build)
    return roof


COMM_COUNTERS = ("two_loops", "two_loop_exchanges", "allreduce_launches", "p2p_exchanges")
COMM_CLASSED = ("timed_exchanges", "exchange_us", "local_wait_us")


def comm_counters_add(acc, before, after):
    """acc += (after - before) over the counters of Context.comm_info()"""
    acc = acc or {**{k: 0 for k in COMM_COUNTERS}, **{k: {"other": 0, "two_loop": 0} for k in COMM_CLASSED}, "exchange_hist": None}
    for k in COMM_COUNTERS:
        acc[k] += after[k] - before[k]
    for k in COMM_CLASSED:
        for c in ("other", "two_loop"):
            acc[k][c] += after[k][c] - before[k][c]
    if after.get("exchange_hist"):  # (ABI 5: exchanges by duration; the test double has none)
        h = acc["exchange_hist"] or {c: [0] * len(after["exchange_hist"][c]) for c in ("other", "two_loop")}
        for c in ("other", "two_loop"):
            h[c] = [x + (y - z) for x, y, z in zip(h[c], after["exchange_hist"][c], before["exchange_hist"][c])]
        acc["exchange_hist"] = h
    return acc


def exchange_figures(ctx, ci_sum, roof):
    """What the communicator of this rank really spans and what one cross-rank exchange costs, over the timed regions
    (lbfgs_hip_ctx_comm_info; `ci_sum`: the counters accumulated over the timed regions).  The figures also go into `roof`
    (roofline.exchange_us_mean, roofline.exchanges_per_two_loop): they are what a scaling shortfall is attributed with."""
    try:
        ci = ctx.comm_info()
    except Exception as e:  # noqa: BLE001
        print(f"[bench] comm_info unavailable: {e}", file=sys.stderr)
        return None
    comm = {k: ci[k] for k in ("kind", "world", "rank", "ranks_seen", "rank_seen", "mailbox_placement", "peers_device",
                               "peers_host", "exclusive_device", "resident_fallbacks")}
    z = ci_sum or comm_counters_add(None, ci, ci)
    d = lambda k: z[k]  # noqa: E731
    dd = lambda k, c: z[k][c]  # noqa: E731
    two_loops, tl_x = d("two_loops"), d("two_loop_exchanges")
    comm["exchanges_per_two_loop"] = (tl_x / two_loops) if two_loops else None
    timed = dd("timed_exchanges", "two_loop")
    if timed:
        comm["exchange_us_mean"] = dd("exchange_us", "two_loop") / timed
        comm["local_wait_us_mean"] = dd("local_wait_us", "two_loop") / timed
        comm["exchange_timing"] = ("measured on the device by the workgroup that closes a reduction across ranks (wall clock): " + (
            "its sums into the slot + flag, until the gate kernel, the ncclAllReduce behind it and the post kernel on the second "
            "stream have answered (gated exchange); exchanges inside two-loops only" if ci["kind"] == "rccl" else
            "stores to every peer's mailbox + wait for every peer's values; exchanges inside two-loops only"))
    elif ci["kind"] == "rccl" and roof.get("per_iteration_ms", {}).get("allreduce_launches"):
        pi = roof["per_iteration_ms"]
        comm["exchange_us_mean"] = pi["allreduce"] / pi["allreduce_launches"] * 1e3
        comm["local_wait_us_mean"] = None
        comm["exchange_timing"] = "HIP events around the grouped ncclAllReduce launches on the compute stream (all of them, not only the two-loop's)"
    else:
        comm["exchange_us_mean"] = comm["local_wait_us_mean"] = None
    # the DISTRIBUTION of the exchanges inside two-loops over the timed regions (device histogram, lbfgs_hip_comm_info ABI 5):
    # one real multi-GPU run then places the leg on profiles/r05_scaling_model.md's latency axis -- its columns are per-exchange
    # latencies L, and a mean hides whether 1 exchange in 100 waited for a straggler
    hist = (z.get("exchange_hist") or {}).get("two_loop")
    if hist and sum(hist) > 0:
        from rust_lbfgs_amd.api import exchange_quantile

        comm["exchange_us_p50"] = exchange_quantile(hist, 0.50)
        comm["exchange_us_p99"] = exchange_quantile(hist, 0.99)
        comm["exchange_hist_counted"] = sum(hist)
    else:
        comm["exchange_us_p50"] = comm["exchange_us_p99"] = None
    # (maxima cannot be differenced: since the context was created, warm-up and start-up self-tests included)
    comm["exchange_us_max_since_start"] = (ci.get("exchange_us_max") or {}).get("two_loop")
    comm["local_wait_us_max_since_start"] = (ci.get("local_wait_us_max") or {}).get("two_loop")
    other = dd("timed_exchanges", "other")
    comm["exchange_us_mean_outside_two_loop"] = (dd("exchange_us", "other") / other) if other else None
    roof["exchange_us_mean"] = comm["exchange_us_mean"]
    roof["exchanges_per_two_loop"] = comm["exchanges_per_two_loop"]
    return comm


def measure(env, ctx, label, vector_free=False, repeats=1):
    """`repeats` times: a fresh state, W warm-up steps (history full), then exactly K timed steps between barriers;
    max over ranks.  Every rank executes the same barriers and collectives even if its own run failed, so a failure
    can never leave a peer waiting."""
    import numpy as np

    import rust_lbfgs_amd as R
    from rust_lbfgs_amd import _ffi, objectives

    a = env.a
    if a.grid:
        ctx.set_grid(a.grid)
    builder = R.lbfgs().with_m(a.m).with_epsilon(0.0)
    if vector_free:
        builder = builder.with_vector_free(True)
    x0 = np.zeros(ctx.n_local)
    hold = {"state": None, "restarts": 0, "vf_fallbacks": 0}
    prefill = max(0, a.m + 2 - a.warmup)  # history must be full (bound = m) before anything is timed

    def fresh():
        if hold["state"] is not None:
            if vector_free:
                hold["vf_fallbacks"] += hold["state"].vector_free_fallbacks()
            hold["state"].close()
            hold["state"] = None
        hold["state"] = builder.build(x0, objectives.Quadratic(fuse_line_eval=a.line_eval), ctx=ctx)

    def step():
        try:
            return hold["state"].propagate()
        except R.LbfgsError as e:
            if e.code <= -100:
                raise  # backend / communicator failure
            # converged to rounding error (the line search cannot make progress): start over
            fresh()
            hold["restarts"] += 1
            return hold["state"].propagate()

    ok = 1.0
    dts, ncalls = [], 0
    ci_sum = None  # the communicator's counters over the TIMED regions only (history full: steady-state exchange counts)
    auto = repeats <= 0
    nrep = 5 if auto else repeats
    rep = -1
    while rep + 1 < nrep:
        rep += 1
        try:
            if ok:
                fresh()
        except R.LbfgsError as e:
            print(f"[bench] rank {env.rank}: {label} failed to build its state: {e}", file=sys.stderr)
            ok = 0.0
        # Every rank has torn its old state down and built the new one before any rank launches a kernel that waits for its
        # peers.  One process per GPU does not need this; ranks that share a PROCESS do (tools/eight_ranks_one_gpu.py): hipFree
        # waits for every stream of its process, the sibling rank's included -- whose kernel may be waiting for this very rank.
        env.barrier(ctx)
        try:
            if ok:
                for _ in range(prefill + a.warmup):
                    step()
                if not a.no_prof and rep == 0:
                    ctx.prof_enable(True)
                    ctx.prof_reset()
                    ctx.prof_enable(False)
                ci_before = ctx.comm_info()
        except R.LbfgsError as e:
            print(f"[bench] rank {env.rank}: {label} failed in warm-up: {e}", file=sys.stderr)
            ok = 0.0
        env.barrier(ctx)
        t0 = time.perf_counter()
        try:
            if ok:
                for i in range(a.steps):
                    if not a.no_prof:
                        ctx.prof_enable(i % max(1, a.prof_every) == 0)
                    ncalls += step().ncall
        except R.LbfgsError as e:
            print(f"[bench] rank {env.rank}: {label} failed in the timed region: {e}", file=sys.stderr)
            ok = 0.0
        env.barrier(ctx)
        dt = time.perf_counter() - t0
        try:
            if ok:
                ci_sum = comm_counters_add(ci_sum, ci_before, ctx.comm_info())
        except R.LbfgsError as e:
            print(f"[bench] rank {env.rank}: comm_info failed: {e}", file=sys.stderr)
        dts.append(env.reduce(dt, "MAX"))
        ok = env.reduce(ok, "MIN")
        if ok != 1.0:
            break
        if auto and rep == 0:
            # as many repeats as it takes to time --min-timed-seconds of iterations (the same number on every rank: dts[0] is
            # the max over ranks), so that the run holds the GPU long enough for an outside observer to see it busy
            nrep = int(min(40, max(5, -(-a.min_timed_seconds // max(dts[0], 1e-6)))))
    res = None
    if ok == 1.0:
        ctx.prof_enable(False)
        n_local = ctx.n_local
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None}
        if not a.no_prof:
            ns, ms_step = ctx.prof_read(_ffi.K_TWOLOOP_STEP)
            nt, ms_all = ctx.prof_read(_ffi.K_TWOLOOP_ALL)
            nu, ms_upd = ctx.prof_read(_ffi.K_UPDATE)
            nv, ms_eval = ctx.prof_read(_ffi.K_EVAL)
            nc, ms_comm = ctx.prof_read(_ffi.K_COMM)
            nr, ms_res = ctx.prof_read(_ffi.K_TWOLOOP_RESIDENT)
            if nr and not vector_free:
                # The running vector (or, for shards larger than the chip, the first n_res elements of it: "hybrid") stays
                # in registers + LDS and the whole recursion is ONE kernel: the on-chip part streams g once, every s and
                # y twice and writes d once -- (4m + 1) passes of 8 bytes per element; the part of q that stays in HBM
                # costs the kernel-per-step path's 8m - 1.
                avg_ms = ms_res / nr
                n_res = min(ctx.resident_elements(), n_local)
                nbytes = 8 * ((4 * a.m + 1) * n_res + (8 * a.m - 1) * (n_local - n_res))
                ach = nbytes / (avg_ms * 1e-3) / 1e9
                roof.update(achieved=ach, frac=ach / HBM_PEAK_GBPS, kernel="two_loop_resident_kernel<ER,NT>",
                            launches=nr, avg_ms=avg_ms, bytes_per_launch=nbytes, resident_elements=n_res,
                            note="algorithmic bytes of THIS kernel: (4m+1) n-vector passes over the elements of q it keeps "
                                 "on the chip, (8m-1) over the rest (hybrid: shards larger than ~1.25e7 elements).  In the hybrid "
                                 "form a 256 MiB slice of the rest keeps the default cache policy and is served by the Infinity "
                                 "Cache between steps: those bytes are requested by the kernel (and counted here and by "
                                 "`traffic`, which counts what leaves the L2s) but do not all reach HBM",
                            infinity_cache_bound=ic_bound(ctx, a.m, n_local, n_res, nbytes, avg_ms))
            elif ns:
                avg_ms = ms_step / ns
                ach = 32.0 * n_local / (avg_ms * 1e-3) / 1e9  # 3 reads + 1 write of f64 per element
                roof.update(achieved=ach, frac=ach / HBM_PEAK_GBPS, kernel="stream_kernel<OpTwoLoopStep<*,false,0>>",
                            launches=ns, avg_ms=avg_ms, bytes_per_launch=32 * n_local)
            if roof["achieved"]:
                # HBM bytes per launch: rocprofv3 PMC counters cannot be collected from inside this process; the
                # figure is taken from the committed counter passes of THIS command (tools/profile_round.sh) when
                # they were made at this shard size and for this kernel, and the record names them -- otherwise null
                roof.update(traffic_lookup(n_local, a.m, roof["kernel"], resident_elements=roof.get("resident_elements")))
            if nt:
                t_tl = ms_all / nt
                # 8*b passes of 8 bytes is the fused minimum that respects the dot->axpy dependency;
                # the first numerator s.(-g) comes out of the history-update kernel, so the exact recursion
                # is charged 8*b - 2 passes (it actually moves 8*b - 1: the last step re-reads g for the next g.d)
                passes = (4 * a.m + 3) if vector_free else (8 * a.m - 2)
                conv_bytes = 8 * passes * n_local
                note = "per GPU: this rank's shard, incl. the all-reduces inside the recursion"
                tl = {"ms": t_tl, "calls": nt, "resident_kernel": bool(nr) and not vector_free}
                if nr and not vector_free:
                    # the persistent kernel IS the recursion: the bytes it moves are roofline.bytes_per_launch; the figure on
                    # the kernel-per-step convention (8m-2 passes) is kept for comparison across shard sizes under a name
                    # that says what it is -- it counts bytes that were never moved and can exceed the HBM peak
                    moved = roof["bytes_per_launch"]
                    tl.update(bytes=moved, algorithmic_GBps=moved / (t_tl * 1e-3) / 1e9,
                              frac=moved / (t_tl * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              GBps_on_kernel_per_step_bytes=conv_bytes / (t_tl * 1e-3) / 1e9,
                              kernel_per_step_bytes=conv_bytes, kernel_per_step_passes=passes,
                              note=note + "; bytes = what the persistent kernel moves (4m+1 passes over the on-chip "
                                          "elements of q, 8m-1 over the rest)")
                else:
                    gbps = conv_bytes / (t_tl * 1e-3) / 1e9
                    tl.update(bytes=conv_bytes, passes=passes, algorithmic_GBps=gbps, frac=gbps / HBM_PEAK_GBPS, note=note)
                roof.update(two_loop=tl)
            sampled = max(nt, 1)  # steps whose kernels were timed (every --prof-every-th step of the timed regions)
            roof["per_iteration_ms"] = {
                "two_loop": ms_all / sampled, "history_update": ms_upd / sampled, "line_eval": ms_eval / sampled,
                "allreduce": ms_comm / sampled, "allreduce_launches": nc / sampled, "sampled_steps": nt}
        comm = exchange_figures(ctx, ci_sum, roof)
        srt = sorted(dts)
        med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        first = prefill + a.warmup + 1  # LbfgsState::propagate calls count from 1 (the first is a no-op: lbfgs.rs:507-510)
        res = dict(label=label, value=a.steps / med, ms_per_step=med / a.steps * 1e3, roofline=roof, n_local=n_local, comm=comm,
                   prefill=prefill, trials=ncalls / max(a.steps * len(dts), 1), restarts=hold["restarts"],
                   repeats=[round(a.steps / d, 3) for d in dts], best=a.steps / srt[0], timed_s=sum(dts),
                   window={"first_iteration": first, "last_iteration": first + a.steps - 1,
                           "line_search_trials_per_step": ncalls / max(a.steps * len(dts), 1),
                           "note": "every repeat times the SAME iterations of a freshly built run (propagate calls, counted from 1); "
                                   "later iterations of this problem need fewer line-search trials and read faster"})
    if hold["state"] is not None:
        try:
            if vector_free:
                hold["vf_fallbacks"] += hold["state"].vector_free_fallbacks()
                if res is not None:
                    res["vector_free_fallbacks"] = hold["vf_fallbacks"]
            hold["state"].close()
        except Exception:  # noqa: BLE001
            pass
    return res


def ic_bound(ctx, m, n_local, n_res, nbytes, avg_ms):
    """Upper bound on the bytes of one hybrid launch that the Infinity Cache can serve instead of HBM, so that the record bounds
    the HBM-channel rate from below by itself: only the slice of the HBM part of q that keeps the default cache policy
    (LBFGS_HIP_RESIDENT_PLAIN_MB, 256 MiB by default; everything else carries `nt`) can be found there, and each of the 2m
    steps after the first reads it once: <= 2m x slice bytes."""
    if n_res >= n_local:
        return None  # all of q on the chip: nothing of it is streamed
    try:
        slice_mb = float(os.environ.get("LBFGS_HIP_RESIDENT_PLAIN_MB", "256"))
    except ValueError:
        slice_mb = 256.0
    slice_bytes = min(slice_mb * 2 ** 20, 8.0 * (n_local - n_res))
    absorbed = 2 * m * slice_bytes
    hbm_min = (nbytes - absorbed) / (avg_ms * 1e-3) / 1e9
    return {"slice_bytes": slice_bytes, "absorbed_bytes_at_most": absorbed, "hbm_GBps_at_least": hbm_min,
            "hbm_frac_at_least": hbm_min / HBM_PEAK_GBPS,
            "says": "at most 2m x the cache slice of q (re-read once per step) can come from the Infinity Cache: the HBM channels "
                    "carry at least (bytes_per_launch - that) / avg_ms"}


def calibrate(ctx, reps=20):
    """What this box's HBM delivers to the plainest kernels of the library: copy (1r 1w) and triad y += c*x (2r 1w), on
    vectors of the bench's shard size.  SURVEY 8(d) asks for the achievable-copy figure beside the 8 TB/s spec peak.
    No reductions here: nothing in this function is a collective, so ranks may run it independently."""
    from rust_lbfgs_amd.math import DeviceVec

    u, v = DeviceVec(ctx), DeviceVec(ctx)
    try:
        out = {}
        if ctx.n_local == 0:
            return {"copy_1r1w_GBps": None, "triad_2r1w_GBps": None}
        for name, fn, passes in (("copy_1r1w_GBps", lambda: v.veccpy(u), 2), ("triad_2r1w_GBps", lambda: v.vecadd(u, 1e-9), 3)):
            for _ in range(3):
                fn()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            ctx.sync()
            dt = (time.perf_counter() - t0) / reps
            out[name] = round(8.0 * passes * ctx.n_local / dt / 1e9, 1)
        return out
    finally:
        u.free()
        v.free()


def compose(a, world, results, ext, legs=None):
    """The one JSON line: the best communicator as `value`, everything measured in `config`."""
    best = max(results, key=lambda r: r["value"])
    cfg = {"workload": f"hashed diagonal quadratic (cond 1e3), n={a.n}, m={a.m}, MoreThuente, crate defaults, "
                       f"x/g/s/y sharded contiguously over {world} GPU(s)",
           "n": a.n, "m": a.m, "n_local_rank0": best["n_local"], "prefill_iters": best["prefill"],
           "line_search_trials_per_step": best["trials"], "restarts": best["restarts"],
           "repeats": len(best["repeats"]), "repeats_iters_per_sec": best["repeats"],
           "best_repeat_iters_per_sec": round(best["best"], 3), "timed_seconds": round(best["timed_s"], 3),
           "timed_window": best["window"],
           "value_is": "median over the repeats of K steps / (max over ranks of the K-step wall time)",
           "allreduce": best["label"],
           "allreduce_measured_iters_per_sec": {r["label"]: round(r["value"], 3) for r in results},
           "extension_vector_free_two_loop": ext}
    if legs is not None:
        cfg["legs"] = legs
    if best.get("comm"):
        cfg["comm_info"] = best["comm"]
        cfg["ranks_seen"] = best["comm"]["ranks_seen"]
    cfg["allreduce_says"] = LEG_SAYS.get(best["label"], best["label"])
    return {
        "metric": METRIC if world == 1 else f"{METRIC}; {world} GPUs, scalars closed by: {LEG_SAYS.get(best['label'], best['label'])}",
        "value": best["value"],
        "unit": "iters/sec",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": best["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": cfg,
        "roofline": best["roofline"],
    }


def worker_main(a):
    """One rank: measures ONE communicator (N = 1: none) and, on rank 0, prints the line.  --_probe: only the context and
    its start-up self-test (a supervisor's phase 1)."""
    # stdout carries exactly ONE JSON line: RCCL, gloo and friends print banners to fd 1, so park it on stderr
    t_start = time.monotonic()
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    comm = a.comm
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env > 1 and comm == "auto":
        comm = "p2p"
    # N = 1: the CPU baseline (~45 s on one core at the metric's size) runs BESIDE the GPU work, in a child pinned to one
    # core that is started first and collected last
    cpu_child = None
    if world_env <= 1 and not a.no_cpu_baseline and not a.probe and "RANK" not in os.environ:
        cpu_child = start_cpu_baseline(a)
    # N = 1 is the only line a one-GPU box ever gives: it exists from the moment measure() returns (`hold["line"]`), everything
    # after that only ADDS to it inside one budget, and a SIGTERM / SIGINT prints the best line so far (as the N > 1 supervisor does)
    single = world_env <= 1 and "RANK" not in os.environ and not a.probe
    hold = {"line": None, "cpu_child": cpu_child, "done": False, "t_post": None, "cuts": []}

    def left():
        """seconds left for what FOLLOWS the measurement: of --post-budget (from the end of the timed measurement) and of
        --total-budget (from the start of the process), less a reserve for composing and printing the line"""
        post = a.post_budget - (time.monotonic() - hold["t_post"]) if hold["t_post"] is not None else a.post_budget
        return min(post, a.total_budget - (time.monotonic() - t_start)) - 3.0

    def emit(line, signum=None):
        if hold["done"] or line is None:
            return False
        hold["done"] = True
        line["config"]["budget"] = {"post_s": a.post_budget, "total_s": a.total_budget, "used_s": round(time.monotonic() - t_start, 1),
                                    "cut": list(hold["cuts"])}
        if signum is not None:
            line["config"]["interrupted_by_signal"] = int(signum)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        return True

    def on_signal(signum, _frame):
        if hold["done"]:
            return
        if Job.proc is not None:  # a rocprofv3 child run: end it with its group
            kill_group(Job.proc)
        line = hold["line"]
        print(f"[bench] signal {signum}: stopping; {'printing the line measured so far' if line else 'nothing measured yet'}", file=sys.stderr)
        if line is not None and not a.no_cpu_baseline and line.get("cpu_baseline", {}).get("value") is None:
            try:  # (only if it has finished: no time to wait now)
                cb = collect_cpu_baseline(hold["cpu_child"], a, timeout=0.5) if hold["cpu_child"] is not None else None
                hold["cpu_child"] = None
                if cb is not None:
                    if cb.get("value") is None:
                        cb["sample"] = f"cut by signal {signum}: " + cb["sample"]
                    line["cpu_baseline"] = cb
            except Exception:  # noqa: BLE001  (e.g. the signal arrived inside the main flow's own communicate())
                pass
        hold["cuts"].append(f"signal {signum}")
        ok = emit(line, signum)
        if hold["cpu_child"] is not None:
            try:
                hold["cpu_child"].kill()
            except OSError:
                pass
        os._exit(0 if ok else 1)

    if single:
        signal.signal(signal.SIGTERM, on_signal)
        signal.signal(signal.SIGINT, on_signal)

    env = Env(a, comm)
    a.gpus = env.world

    import rust_lbfgs_amd  # noqa: F401  (fails loudly if the HIP extension is not built)

    results, ext = [], {}
    t_ctx = time.perf_counter()
    ctx, label = make_context(env, comm if env.world > 1 else "none")
    if a.probe:
        placement = getattr(ctx, "p2p_placement", None) if ctx is not None else None
        if ctx is not None:
            ctx.close()
        env.finish()
        if env.rank == 0 and ctx is not None:
            os.write(real_stdout, (json.dumps({"probe": "ok", "comm": label, "mailboxes": placement,
                                               "seconds": round(time.perf_counter() - t_ctx, 2)}) + "\n").encode())
        return 0 if ctx is not None else 4

    def recompose():
        """the contract line from everything measured so far (rank 0)"""
        if env.rank != 0 or not results:
            return
        out = compose(a, env.world, results, ext)
        if results[0].get("mailboxes"):
            out["config"]["p2p_mailboxes"] = results[0]["mailboxes"]
        if env.world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = (hold["line"] or {}).get("cpu_baseline") or cpu_baseline_missing(
                "not collected yet: the baseline child runs beside the GPU work and is collected last")
        hold["line"] = out

    if ctx is not None:
        r = measure(env, ctx, label, repeats=a.repeats)
        hold["t_post"] = time.monotonic()
        if r is not None:
            results.append(r)
            recompose()  # the line exists from here on
            if r["roofline"].get("achieved"):
                try:  # both denominators: the spec peak (frac) and what a plain copy achieves on this box
                    cal = calibrate(ctx)
                    r["roofline"]["calibration"] = cal
                    if cal["copy_1r1w_GBps"]:
                        r["roofline"]["frac_of_copy"] = r["roofline"]["achieved"] / cal["copy_1r1w_GBps"]
                except Exception as e:  # noqa: BLE001
                    print(f"[bench] calibration skipped: {e}", file=sys.stderr)
            if getattr(ctx, "p2p_placement", None):
                r["mailboxes"] = ctx.p2p_placement
            recompose()
            if not a.no_vector_free and a.m <= 10 and single and left() < 90.0:
                hold["cuts"].append("vector-free extension not measured: budget")
            elif not a.no_vector_free and a.m <= 10:
                # EXTENSION, reported beside the headline, never as `value`: the same iteration with the
                # two-loop carried out in Gram-coefficient space (4m+3 passes, 2 all-reduces)
                try:  # (an extra: whatever happens to it, the line of the measurement above stands)
                    rv = measure(env, ctx, label + "+vector_free", vector_free=True, repeats=3 if a.repeats <= 0 else min(a.repeats, 3))
                except Exception as e:  # noqa: BLE001
                    if env.world > 1:
                        raise  # (ranks must fail together: the supervisor files the leg)
                    print(f"[bench] the vector-free extension's measurement failed: {type(e).__name__}: {e}", file=sys.stderr)
                    hold["cuts"].append(f"vector-free extension not measured: {type(e).__name__}: {e}")
                    rv = None
                if rv is not None:
                    tl = rv["roofline"].get("two_loop", {})
                    ext[label] = {"iters_per_sec": round(rv["value"], 3), "two_loop_ms": tl.get("ms"),
                                  "fallbacks": rv.get("vector_free_fallbacks"),
                                  "fallbacks_note": "iterations (warm-up included) whose coefficient-space direction failed its "
                                                    "run-time ||d||^2 check and was redone by the exact recursion",
                                  "two_loop_passes": 4 * a.m + 3, "repeats_iters_per_sec": rv["repeats"]}
                    recompose()
        try:
            ctx.close()
        except Exception as e:  # noqa: BLE001  (the measurement is in hand)
            print(f"[bench] closing the context failed: {type(e).__name__}: {e}", file=sys.stderr)
        if results and results[0]["roofline"].get("achieved") and live_traffic_wanted(a):
            roof = results[0]["roofline"]
            try:  # (after ctx.close(): the children get the whole GPU)
                apply_live_traffic(roof, live_traffic(a, roof, timeout_s=a.rocprof_timeout, left=left))
            except Exception as e:  # noqa: BLE001  (the committed look-up stays)
                roof["traffic_live_error"] = f"{type(e).__name__}: {e}"
            recompose()
            try:
                roof.update(live_kernel_time(a, roof, timeout_s=a.rocprof_timeout, left=left))
            except Exception as e:  # noqa: BLE001
                roof["rocprofv3_error"] = f"{type(e).__name__}: {e}"
            for k in ("traffic_live_error", "rocprofv3_error"):
                if roof.get(k):
                    hold["cuts"].append(f"{k}: {roof[k]}")
            recompose()

    rc = 0
    if env.rank == 0:
        if not results:
            print("bench.py: the communicator produced no result", file=sys.stderr)
            rc = 3
        elif env.world == 1 and not a.no_cpu_baseline:
            # (the Popen stays in `hold` while communicate() waits: a signal that arrives meanwhile must still find the child)
            cb = collect_cpu_baseline(hold["cpu_child"], a, timeout=max(0.5, left()) if single else 420.0)
            hold["cpu_child"] = None
            if cb.get("value") is None:
                hold["cuts"].append("cpu_baseline: " + cb["sample"])
            hold["line"]["cpu_baseline"] = cb
    if hold["cpu_child"] is not None:
        hold["cpu_child"].kill()
        hold["cpu_child"] = None
    env.finish()
    emit(hold["line"])
    return rc


# ======================================================================================== supervisor (N > 1)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


LEG_COMM = {"p2p": "p2p", "p2p-per-step": "p2p", "p2p-host": "p2p-host", "rccl": "rccl", "rccl-per-step": "rccl", "callback": "callback"}
# what closes the scalar reductions of a leg, in words: an N > 1 line says it in `metric` and `config.allreduce`, so that nobody
# reads a number of the in-kernel exchange as a number of RCCL (BASELINE.json's north star names "a scalar RCCL all-reduce")
LEG_SAYS = {
    "p2p": "p2p in-kernel exchange (tagged stores into device mailboxes mapped over xGMI, inside the reducing kernels; the two-loop "
           "is one persistent kernel) -- NOT RCCL",
    "p2p-per-step": "p2p in-kernel exchange (device mailboxes over xGMI), one kernel per two-loop step -- NOT RCCL",
    "p2p-host": "p2p in-kernel exchange through host-coherent mailboxes (reached over PCIe) -- NOT RCCL",
    "rccl": "rccl ncclAllReduce(ncclDouble, ncclSum), one per reduction; inside the two-loop the all-reduces run on a second stream "
            "behind gate kernels while the recursion stays ONE persistent kernel (gated exchange)",
    "rccl-per-step": "rccl ncclAllReduce(ncclDouble, ncclSum) on the compute stream, one per reduction; the two-loop is one kernel per step",
    "callback": "host-staged all-reduce through a host callback (torch.distributed/gloo in this program's own legs; last resort) -- NOT RCCL",
    "none": "single GPU: no communicator",
}


def ran_as(leg, line):
    """The leg a measurement really took.  "rccl" and "p2p" ASK for the persistent two-loop kernel; the library decides -- e.g.
    lbfgs_hip_ctx_create's trial of the gated exchange failed on some rank (all ranks then run a kernel per step), or the shard is
    not eligible -- and the record shows it: roofline.two_loop.resident_kernel."""
    tl = (line.get("roofline") or {}).get("two_loop") or {}
    if leg in ("rccl", "p2p") and tl.get("resident_kernel") is False:
        return leg + "-per-step"
    return leg


def rccl_beside(report, lines):
    """config.rccl: the RCCL leg at the top level of `config` whatever leg `value` comes from -- its iterations/sec, its
    two-loop and what one ncclAllReduce cost (HIP events around the launches), or why there is no such figure."""
    rep = report.get("rccl")
    out = {"iters_per_sec": None, "two_loop_ms": None, "allreduce_us_mean": None, "allreduces_per_two_loop": None,
           "ranks_seen": None, "status": "not run" if rep is None else rep.get("status"),
           "says": LEG_SAYS["rccl"]}
    for lg, j in lines:
        if lg not in ("rccl", "rccl-per-step") or (lg == "rccl-per-step" and out["iters_per_sec"] is not None):
            continue
        out["says"] = LEG_SAYS[lg]
        out["status"] = (report.get(lg) or {}).get("status", out["status"])
        roof = j.get("roofline") or {}
        ci = j["config"].get("comm_info") or {}
        out.update(iters_per_sec=round(j["value"], 3), two_loop_ms=(roof.get("two_loop") or {}).get("ms"),
                   allreduce_us_mean=ci.get("exchange_us_mean"), allreduces_per_two_loop=ci.get("exchanges_per_two_loop"),
                   ranks_seen=ci.get("ranks_seen"))
    return out


def passthrough(a, leg, probe=False, vector_free=True):
    args = ["--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup), "--dim", str(a.n), "--hist",
            str(a.m), "--repeats", str(a.repeats), "--min-timed-seconds", str(a.min_timed_seconds), "--prof-every",
            str(a.prof_every), "--line-eval", str(a.line_eval), "--pg-backend", a.pg_backend, "--comm", LEG_COMM[leg],
            "--no-cpu-baseline", "--_rank-mode"]
    if probe:
        args.append("--_probe")
    if a.no_prof:
        args.append("--no-prof")
    if a.no_vector_free or not vector_free:
        args.append("--no-vector-free")
    if a.grid:
        args += ["--grid", str(a.grid)]
    if a.device >= 0:
        args += ["--device", str(a.device)]
    if a.exclusive_device >= 0:
        args += ["--exclusive-device", str(a.exclusive_device)]
    return args


class Job:
    """The child job that is running right now (so that a signal handler can end it)."""
    proc = None


def kill_group(p):
    import signal

    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(p.pid, sig)
        except (ProcessLookupError, PermissionError):
            break
        try:
            p.wait(timeout=5)
            break
        except subprocess.TimeoutExpired:
            continue


def run_child(cmd, env, timeout):
    """-> (status, stdout).  The child gets its own process group, so a hung job is killed with all its descendants
    (torch.distributed.run's workers included); never a re-exec of this process."""
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True)
    Job.proc = p
    try:
        out, _ = p.communicate(timeout=max(1.0, timeout))
        return ("ok" if p.returncode == 0 else f"exit code {p.returncode}"), out
    except subprocess.TimeoutExpired:
        kill_group(p)
        try:
            out, _ = p.communicate(timeout=5)
        except Exception:  # noqa: BLE001
            out = ""
        return f"timed out after {timeout:.0f} s (killed)", out or ""
    finally:
        Job.proc = None


def last_json(text):
    for ln in reversed((text or "").splitlines()):
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def supervisor_main(a):
    """N > 1.  This process never touches a GPU: it runs child jobs -- a probe per communicator, then a measurement per
    communicator that passed -- each under a timeout that is its share of what is left of ONE total budget."""
    import signal

    t_start = time.monotonic()
    left = lambda: a.total_budget - (time.monotonic() - t_start)  # noqa: E731
    # stdout carries exactly ONE JSON line (gloo prints connection banners to fd 1): park it on stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    under_launcher = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ["WORLD_SIZE"]) if under_launcher else a.gpus
    a.gpus = world
    probes, report, lines = {}, {}, []
    state = {"done": False, "cpu_child": None}
    # The CPU baseline (the reference is single-threaded: one pinned core, the N = 1 configuration) runs BESIDE the legs: this
    # process touches no GPU and would otherwise idle for the whole budget.  Rank 0's supervisor only.
    if rank == 0 and not a.no_cpu_baseline:
        state["cpu_child"] = start_cpu_baseline(a)

    def final_line():
        """the best measurement so far as the run's line (None if there is none)"""
        if not lines:
            return None
        leg, best = max(lines, key=lambda t: t[1]["value"])
        best = json.loads(json.dumps(best))
        cfg = best["config"]
        cfg["probes"], cfg["legs"] = probes, report
        cfg["allreduce"] = leg
        cfg["allreduce_says"] = LEG_SAYS.get(leg, leg)
        cfg["value_from_leg"] = leg
        cfg["rccl"] = rccl_beside(report, lines)
        best["metric"] = f"{METRIC}; {world} GPUs, scalars closed by: {LEG_SAYS.get(leg, leg)}"
        cfg["allreduce_measured_iters_per_sec"] = {lg: round(j["value"], 3) for lg, j in lines}
        ext = {}
        for lg, j in lines:
            for k, v in (j["config"].get("extension_vector_free_two_loop") or {}).items():
                ext[lg] = v
        cfg["extension_vector_free_two_loop"] = ext
        cfg["launch"] = ("torch.distributed.run ranks as supervisors, one child rank each per job"
                         if under_launcher else "self-launched: one torch.distributed.run child per job")
        cfg["budget"] = {"total_s": a.total_budget, "used_s": round(time.monotonic() - t_start, 1)}
        cfg["ranks_seen"] = (cfg.get("comm_info") or {}).get("ranks_seen")
        return best

    def attach_cpu_baseline(best, wait_s):
        """the baseline child's result into the line (it has had the whole run to finish; `wait_s` more at most)"""
        child = state["cpu_child"]
        if child is None or best is None:
            return
        # (the Popen stays in `state` while communicate() waits -- up to left() + 60 s: a signal that arrives meanwhile must
        # still find the child, ~22 GB of host memory and minutes of CPU time, and kill it)
        cb = collect_cpu_baseline(child, a, timeout=max(0.5, wait_s))
        state["cpu_child"] = None
        cb["where"] = ("rank 0's supervisor process (no GPU), started before the first probe and timed while the legs ran; the "
                       "reference is single-threaded, so this is the N = 1 figure whatever --gpus says")
        best["cpu_baseline"] = cb

    def on_signal(signum, _frame):
        # the driver (or torch.distributed.run, on its behalf) wants this run to end NOW: end the running job and hand over
        # whatever has been measured -- a line from the legs that finished is worth more than none
        if state["done"]:
            return
        state["done"] = True
        pgid = Job.proc.pid if Job.proc is not None else None
        if Job.proc is not None:
            kill_group(Job.proc)
        rc = 1
        if rank == 0:
            best = final_line()
            print(f"[bench] signal {signum}: stopping with {len(lines)} measured leg(s); running job's process group: {pgid}",
                  file=sys.stderr)
            if best is not None:
                best["config"]["interrupted_by_signal"] = int(signum)
                try:
                    attach_cpu_baseline(best, 0.5)  # (only if it has finished: no time to wait now)
                except Exception:  # noqa: BLE001  (e.g. the signal arrived inside the main flow's own communicate())
                    pass
                os.write(real_stdout, (json.dumps(best) + "\n").encode())
                rc = 0
        if state["cpu_child"] is not None:
            try:
                state["cpu_child"].kill()
            except OSError:
                pass
        os._exit(rc)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)

    dist = None
    if under_launcher:  # the ranks' supervisors coordinate over gloo (CPU only)
        import datetime

        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        import torch.distributed as dist

        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=a.total_budget + 120))
    else:
        try:  # page the framework in BEFORE the first child is on the clock (a fresh box takes a minute over it; CPU only)
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001
            pass
    base_env = dict(os.environ)
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
    base_env.setdefault("GLOO_SOCKET_IFNAME", "lo")         # one node: never depend on the hostname resolving

    def run_job(leg, probe, timeout, vector_free=True):
        """One child job on every rank.  -> (status, json or None); identical status on every rank."""
        # test hooks: "hang*" = a measurement that never returns (its probe passes), "hangprobe" = a probe that never returns,
        # "failprobe" = a probe that fails at once
        hook = leg.startswith("hang") or leg == "failprobe"   # ("hang", "hang2", ...: several hung legs in one run)
        if hook and probe and leg != "hangprobe" and leg != "failprobe":
            return "ok", {"probe": "ok", "comm": leg}
        if leg == "failprobe":
            return "exit code 4", None
        leg_env = dict(base_env)
        if leg == "p2p-per-step":  # the p2p communicator with one kernel per two-loop step (no persistent kernel)
            leg_env["LBFGS_HIP_RESIDENT"] = "0"
        if leg == "rccl":  # OPT IN to the gated exchange (the library's default is a kernel per step until the gated form has run with
            leg_env["LBFGS_HIP_RCCL_RESIDENT"] = "1"  # peers): this job is a child with a timeout, and ran_as() files what really ran
        if leg == "rccl-per-step":  # RCCL with one kernel per two-loop step (no gated exchange under the persistent kernel)
            leg_env["LBFGS_HIP_RCCL_RESIDENT"] = "0"
        if hook:
            cmd_tail, child = None, [sys.executable, "-c", "import time; time.sleep(100000)"]
        else:
            # (LBFGS_BENCH_WORKER: the CPU suite runs the rank processes on the test double of the C-ABI)
            cmd_tail = [os.environ.get("LBFGS_BENCH_WORKER") or os.path.join(ROOT, "bench.py")] + passthrough(a, leg, probe, vector_free)
            child = None
        if under_launcher:
            port = decided(free_port() if rank == 0 else None)
            env = dict(leg_env, MASTER_PORT=str(port))
            for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
                      "TORCHELASTIC_USE_AGENT_STORE"):
                env.pop(k, None)  # the child does its own env:// rendezvous on the fresh port
            cmd = child or [sys.executable] + cmd_tail
        else:
            env = leg_env
            cmd = child or [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                            "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + cmd_tail
        status, out = run_child(cmd, env, timeout)
        if under_launcher:  # a job counts only if every rank's child ended well
            all_status = [None] * world
            dist.all_gather_object(all_status, status)
            bad = [f"rank {r}: {st}" for r, st in enumerate(all_status) if st != "ok"]
            if bad:
                status = "; ".join(bad)
        j = last_json(out) if (rank == 0 and status == "ok") else None
        if rank == 0 and status == "ok" and j is None:
            status = "no JSON line"
        return decided(status), j

    RESERVE = 8.0  # seconds kept for composing and printing the line

    def decided(value):
        """rank 0's value on every rank: every decision that shapes the sequence of jobs is rank 0's (its clock, its results)"""
        if not under_launcher:
            return value
        box = [value if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    # ---- phase 1: probes
    if os.environ.get("LBFGS_BENCH_LEGS"):  # testing: e.g. "hang,p2p" (a leg that never returns)
        comms = os.environ["LBFGS_BENCH_LEGS"].split(",")
    elif a.comm == "auto":
        comms = ["p2p", "p2p-host", "rccl"]
    else:
        comms = [a.comm]
    passed = []
    comms = list(comms)
    for i, leg in enumerate(comms):  # (the list may grow: a failed "rccl" probe appends "rccl-per-step")
        t0 = time.monotonic()
        # (the first probe also pays for paging the libraries in; RCCL's first communicator over 8 GPUs takes its time)
        tmo = decided(min(a.probe_timeout * (2.0 if (i == 0 or leg == "rccl") else 1.0), left() - RESERVE))
        if tmo < 3.0:
            probes[leg] = {"status": "skipped: budget spent", "seconds": 0.0}
            continue
        status, j = run_job(leg, True, tmo)
        probes[leg] = {"status": status, "seconds": round(time.monotonic() - t0, 1), "timeout_s": round(tmo, 1)}
        if j and j.get("mailboxes"):
            probes[leg]["mailboxes"] = j["mailboxes"]
        if status == "ok":
            passed.append(leg)
        elif leg == "rccl" and "rccl-per-step" not in comms:
            # the gated exchange (the persistent kernel served by ncclAllReduce on a second stream) is what no single GPU can try
            # with peers: should RCCL itself be fine and only that form fail, its plain form -- a kernel per step -- is probed too
            comms.append("rccl-per-step")
        if rank == 0:
            print(f"[bench] probe {leg}: {probes[leg]}  ({left():.0f} s of the budget left)", file=sys.stderr)

    # ---- phase 2: measurements.  Order: p2p, (p2p-per-step only if the p2p MEASUREMENT failed), then -- fallback order while
    # there is no result -- p2p-host before rccl; once there is a result the others follow as comparisons, rccl first.
    def measure_leg(leg, legs_after):
        t0 = time.monotonic()
        cap = min(a.leg_timeout, 150.0) if leg.startswith("p2p") else a.leg_timeout
        tmo = decided(min(cap, (left() - RESERVE) / (1 + legs_after)))
        if tmo < 10.0:
            report[leg] = {"status": "skipped: budget spent", "seconds": 0.0, "iters_per_sec": None}
            if rank == 0:
                print(f"[bench] leg {leg}: {report[leg]}", file=sys.stderr)
            return False
        # the vector-free extension is an extra: not once less than a third of the budget is left
        vf = decided(left() >= a.total_budget / 3.0)
        status, j = run_job(leg, False, tmo, vector_free=vf)
        report[leg] = {"status": status, "seconds": round(time.monotonic() - t0, 1), "timeout_s": round(tmo, 1),
                       "iters_per_sec": round(j["value"], 3) if j else None}
        if j:  # what the leg's communicator really spanned and what an exchange cost it (lbfgs_hip_ctx_comm_info on rank 0)
            ci = j["config"].get("comm_info") or {}
            report[leg].update(ranks_seen=ci.get("ranks_seen"), mailboxes_mapped=(ci.get("peers_device"), ci.get("peers_host")),
                               mailbox_placement=ci.get("mailbox_placement"), exchange_us_mean=ci.get("exchange_us_mean"),
                               exchange_us_p50=ci.get("exchange_us_p50"), exchange_us_p99=ci.get("exchange_us_p99"),
                               exchange_us_max=ci.get("exchange_us_max_since_start"),
                               local_wait_us_mean=ci.get("local_wait_us_mean"), local_wait_us_max=ci.get("local_wait_us_max_since_start"),
                               exchanges_per_two_loop=ci.get("exchanges_per_two_loop"),
                               two_loop_ms=((j.get("roofline") or {}).get("two_loop") or {}).get("ms"))
        if j:
            ran = ran_as(leg, j)  # (the library may have taken the kernel-per-step form by itself: the line must say what RAN)
            if ran != leg:
                report[leg]["ran_as"] = ran
            lines.append((ran, j))
        if rank == 0:
            print(f"[bench] leg {leg}: {report[leg]}  ({left():.0f} s of the budget left)", file=sys.stderr)
        return j is not None

    def sync_have():
        return decided(len(lines) > 0)

    todo = [c for c in passed]
    explicit = bool(os.environ.get("LBFGS_BENCH_LEGS")) or a.comm != "auto"
    if explicit:
        for i, leg in enumerate(todo):
            measure_leg(leg, len(todo) - 1 - i)
    else:
        if "p2p" in todo:
            others = [c for c in todo if c != "p2p"]
            measure_leg("p2p", len(others))
            if not sync_have():
                measure_leg("p2p-per-step", len(others))
        rest = [c for c in todo if c != "p2p"]
        if not sync_have():  # fallback order
            rest.sort(key=lambda c: {"p2p-host": 0, "rccl": 1, "rccl-per-step": 1}.get(c, 2))
        else:                # comparisons
            rest.sort(key=lambda c: {"rccl": 0, "rccl-per-step": 0, "p2p-host": 1}.get(c, 2))
        for i, leg in enumerate(rest):
            got = measure_leg(leg, len(rest) - 1 - i)
            if leg == "rccl" and not decided(got) and "rccl-per-step" not in rest:  # the gated form gave nothing: RCCL's plain form
                measure_leg("rccl-per-step", len(rest) - 1 - i)
    if not sync_have() and (a.comm == "auto") and "callback" not in report:
        # last resort: the host-staged all-reduce through gloo, only if nothing has produced a result
        measure_leg("callback", 0)

    state["done"] = True
    rc = 0
    if rank == 0:
        best = final_line()
        if best is None:
            print("bench.py: no communicator leg produced a result: " + json.dumps({"probes": probes, "legs": report}), file=sys.stderr)
            rc = 1
        else:
            attach_cpu_baseline(best, left() + 60.0)  # (normally long finished; the driver's limit is 80 s past the budget)
            os.write(real_stdout, (json.dumps(best) + "\n").encode())
    if state["cpu_child"] is not None:
        state["cpu_child"].kill()
        state["cpu_child"] = None
    if dist is not None:
        ok = [rc]
        dist.broadcast_object_list(ok, src=0)
        rc = ok[0]
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main():
    a = parse()
    if a.cpu_child:
        return cpu_baseline_child_main(a)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if a.rank_mode or (a.gpus <= 1 and world_env <= 1):
        return worker_main(a)
    return supervisor_main(a)


if __name__ == "__main__":
    sys.exit(main())
