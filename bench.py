#!/usr/bin/env python3
"""bench.py -- L-BFGS iterations/sec and two-loop HBM GB/s on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dim 100000000] [--hist 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE L-BFGS iteration (LbfgsState::propagate, reference src/lbfgs.rs:503-560) of the
synthetic diagonal quadratic of BASELINE.json config 4: n = 1e8, m = 10, More-Thuente, f64, the
crate's default parameters, everything resident in HBM (device objective, no PCIe in the timed
region).  One step = the line search (per trial one pass over xp and d that returns f and g.d), the
history update (which also forms the accepted x and g) and the fused two-loop recursion.  With N > 1 the n-vector is sharded contiguously over the
ranks (total work fixed => "strong" scaling) and every reduction is closed by an all-reduce of its
f64 scalars: RCCL ncclAllReduce on the compute stream, or the direct xGMI mailbox exchange ("p2p").
With --comm auto (default) both are measured -- RCCL first, then p2p if its start-up self-test
passes on every rank -- and the faster one is reported as `value`; both are listed in `config`.

The JSON line carries, besides the contract fields:
  roofline      the dominant kernel (two-loop step  q += c*u ; out = v.q,  3 reads + 1 write of
                an n-vector = 32 bytes/element algorithmic) timed with HIP events on the launch
                stream over the timed region, against the 8 TB/s HBM peak;
  cpu_baseline  the CPU oracle (reference operation order, 1 thread) on a bounded sample of the
                same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--dim", dest="n", type=int, default=100_000_000, help="n, the number of variables")
    ap.add_argument("--hist", dest="m", type=int, default=10, help="m, the number of L-BFGS corrections")
    ap.add_argument("--no-prof", action="store_true", help="do not time kernels with HIP events in the timed region")
    ap.add_argument("--prof-every", type=int, default=5,
                    help="time kernels with HIP events on every k-th step of the timed region only: an event pair per "
                         "kernel costs ~1.5 us of stream time, 11 %% of an iteration at 100 MB shards when always on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=20_000_000, help="sample size of the CPU baseline (~10 s on one core)")
    ap.add_argument("--grid", type=int, default=0, help="workgroups per launch (0 = library default)")
    ap.add_argument("--comm", default="auto", choices=["auto", "rccl", "p2p", "callback"],
                    help="N>1: how scalars are all-reduced (auto = measure RCCL, then p2p if its self-test passes)")
    ap.add_argument("--pg-backend", default="nccl", help="torch.distributed backend used for rendezvous/barriers")
    ap.add_argument("--no-vector-free", action="store_true",
                    help="skip the extra measurement of the vector-free (Gram) two-loop extension")
    ap.add_argument("--line-eval", type=int, default=2, choices=[0, 1, 2],
                    help="lbfgs_evaluator.fuse_line_eval: 2 = trials write no vectors (default), 1 = every trial writes x and g, "
                         "0 = separate passes")
    ap.add_argument("--device", type=int, default=-1, help="force a device index (testing: several ranks on one GPU)")
    return ap.parse_args()


def cpu_baseline(n_sample, m, n_full):
    """The oracle (C restatement of the reference's sequential arithmetic) on one host core."""
    import numpy as np

    from oracle import oracle as O

    x = np.zeros(n_sample)
    st = O.lbfgs().with_m(m).with_epsilon(0.0).build(x, O.quadratic())
    warm, timed = m + 2, 6
    for _ in range(warm):
        st.propagate()
    t0 = time.perf_counter()
    for _ in range(timed):
        st.propagate()
    dt = time.perf_counter() - t0
    st.close()
    ips_sample = timed / dt
    return {
        "value": ips_sample * n_sample / n_full,
        "unit": "iters/sec",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle (gcc -O2 -ffp-contract=off, sequential sums) on the same quadratic at n={n_sample}, m={m}: "
                  f"{timed} iterations after {warm} warm-up = {ips_sample:.3f} iters/sec, scaled by n_sample/n "
                  f"(every pass is O(n)) to n={n_full}",
    }


class Env:
    """rank / world / torch.distributed plumbing (only rendezvous, barriers and a max over ranks)."""

    def __init__(self, a):
        self.a = a
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dev = self.local_rank if a.device < 0 else a.device
        self.dist = self.torch = None
        if self.world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU
            import torch
            import torch.distributed as dist

            self.torch, self.dist = torch, dist
            torch.cuda.set_device(self.dev)
            if a.pg_backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.dev))
            else:
                dist.init_process_group(a.pg_backend)

    def _tensor(self, v):
        return self.torch.tensor([v], dtype=self.torch.float64, device="cuda" if self.a.pg_backend == "nccl" else "cpu")

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
        if env.dist is not None and os.environ.get("LBFGS_FORCE_RCCL") == "1":
            return sharded_context(a.n, device=env.dev, kind="rccl"), "rccl"
        return R.Context(a.n, device=env.dev), "none"
    ok, ctx = 1.0, None
    try:
        ctx = sharded_context(a.n, device=env.dev, kind=kind)
        if kind == "p2p":  # known-answer reductions through the real code path before trusting it
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
    except Exception as e:  # noqa: BLE001
        print(f"[bench] rank {env.rank}: {kind} communicator unavailable: {e}", file=sys.stderr)
        ok = 0.0
    if env.reduce(ok, "MIN") != 1.0:
        if ctx is not None:
            ctx.close()
        return None, kind
    return ctx, kind


def measure(env, ctx, label, vector_free=False):
    """W warm-up steps, then exactly K timed steps between barriers; max over ranks.  Every rank executes the
    same barriers even if its own run failed, so a failure can never leave a peer waiting."""
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
    hold = {"state": None, "restarts": 0}
    ok = 1.0
    ncalls = 0
    prefill = max(0, a.m + 2 - a.warmup)  # history must be full (bound = m) before anything is timed

    def step():
        try:
            return hold["state"].propagate()
        except R.LbfgsError as e:
            if e.code <= -100:
                raise  # backend / communicator failure
            # converged to rounding error (the line search cannot make progress): start over
            hold["state"].close()
            hold["state"] = builder.build(x0, objectives.Quadratic(fuse_line_eval=a.line_eval), ctx=ctx)
            hold["restarts"] += 1
            return hold["state"].propagate()

    try:
        hold["state"] = builder.build(x0, objectives.Quadratic(fuse_line_eval=a.line_eval), ctx=ctx)
        for _ in range(prefill + a.warmup):
            step()
        if not a.no_prof:
            ctx.prof_enable(True)
            ctx.prof_reset()
            ctx.prof_enable(False)
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
    dt = env.reduce(dt, "MAX")
    ok = env.reduce(ok, "MIN")
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
            if ns:
                avg_ms = ms_step / ns
                ach = 32.0 * n_local / (avg_ms * 1e-3) / 1e9  # 3 reads + 1 write of f64 per element
                roof.update(achieved=ach, frac=ach / HBM_PEAK_GBPS, kernel="stream_kernel<OpTwoLoopStep<*,false,0>>",
                            launches=ns, avg_ms=avg_ms, bytes_per_launch=32 * n_local)
                # HBM bytes per launch from the committed rocprofv3 PMC passes (cannot be collected from inside)
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                    if pm["n_local"] == n_local:
                        roof["traffic"] = pm["traffic_bytes_per_launch"] / 1e9
                        roof["traffic_unit"] = "GB per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_bench_n1e8_m10.md)"
                except Exception:  # noqa: BLE001
                    pass
            if nt:
                t_tl = ms_all / nt
                # 8*b passes of 8 bytes is the fused minimum that respects the dot->axpy dependency;
                # the first numerator s.(-g) now comes out of the history-update kernel, so the exact recursion
                # is charged 8*b - 2 passes (it actually moves 8*b - 1: the last step re-reads g for the next g.d)
                passes = (4 * a.m + 3) if vector_free else (8 * a.m - 2)
                gbps = 8.0 * passes * n_local / (t_tl * 1e-3) / 1e9
                roof.update(two_loop={"ms": t_tl, "algorithmic_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS,
                                      "bytes": 8 * passes * n_local, "passes": passes, "calls": nt,
                                      "note": "per GPU: this rank's shard, incl. the all-reduces inside the recursion"})
            sampled = max(nt, 1)  # steps whose kernels were timed (every --prof-every-th step of the timed region)
            roof["per_iteration_ms"] = {
                "two_loop": ms_all / sampled, "history_update": ms_upd / sampled, "line_eval": ms_eval / sampled,
                "allreduce": ms_comm / sampled, "allreduce_launches": nc / sampled, "sampled_steps": nt}
        res = dict(label=label, value=a.steps / dt, ms_per_step=dt / a.steps * 1e3, roofline=roof, n_local=n_local,
                   prefill=prefill, trials=ncalls / max(a.steps, 1), restarts=hold["restarts"])
    if hold["state"] is not None:
        try:
            hold["state"].close()
        except Exception:  # noqa: BLE001
            pass
    return res


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


def main():
    a = parse()
    # stdout carries exactly ONE JSON line: RCCL, gloo and friends print banners to fd 1, so park it on stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    env = Env(a)
    if env.world != a.gpus:
        if env.world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = env.world

    import rust_lbfgs_amd  # noqa: F401  (fails loudly if the HIP extension is not built)

    kinds = ["rccl", "p2p"] if (env.world > 1 and a.comm == "auto") else [a.comm if env.world > 1 else "none"]
    results, ext = [], {}
    for kind in kinds:
        ctx, label = make_context(env, kind)
        if ctx is None:
            continue
        r = measure(env, ctx, label)
        if r is not None:
            if r["roofline"].get("achieved"):
                try:  # both denominators: the spec peak (frac) and what a plain copy achieves on this box
                    cal = calibrate(ctx)
                    r["roofline"]["calibration"] = cal
                    if cal["copy_1r1w_GBps"]:
                        r["roofline"]["frac_of_copy"] = r["roofline"]["achieved"] / cal["copy_1r1w_GBps"]
                except Exception as e:  # noqa: BLE001
                    print(f"[bench] calibration skipped: {e}", file=sys.stderr)
            results.append(r)
            if not a.no_vector_free and a.m <= 10:
                # EXTENSION, reported beside the headline, never as `value`: the same iteration with the
                # two-loop carried out in Gram-coefficient space (4m+3 passes, 2 all-reduces)
                rv = measure(env, ctx, label + "+vector_free", vector_free=True)
                if rv is not None:
                    tl = rv["roofline"].get("two_loop", {})
                    ext[label] = {"iters_per_sec": round(rv["value"], 3), "two_loop_ms": tl.get("ms"),
                                  "two_loop_passes": 4 * a.m + 3}
        ctx.close()

    out = None
    if env.rank == 0:
        if not results:
            sys.exit("bench.py: no communicator produced a result")
        best = max(results, key=lambda r: r["value"])
        out = {
            "metric": "L-BFGS iters/sec (two-loop HBM GB/s in roofline) at n=1e8, m=10",
            "value": best["value"],
            "unit": "iters/sec",
            "n_gpus": env.world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": best["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"hashed diagonal quadratic (cond 1e3), n={a.n}, m={a.m}, MoreThuente, crate defaults, "
                                   f"x/g/s/y sharded contiguously over {env.world} GPU(s)",
                       "n": a.n, "m": a.m, "n_local_rank0": best["n_local"], "prefill_iters": best["prefill"],
                       "line_search_trials_per_step": best["trials"], "restarts": best["restarts"],
                       "allreduce": best["label"],
                       "allreduce_measured_iters_per_sec": {r["label"]: round(r["value"], 3) for r in results},
                       "extension_vector_free_two_loop": ext},
            "roofline": best["roofline"],
        }
        if env.world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_n, a.m, a.n)
    env.finish()
    if out is not None:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
