#!/usr/bin/env python3
"""bench.py -- L-BFGS iterations/sec and two-loop HBM GB/s on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 100000000] [--m 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE L-BFGS iteration (LbfgsState::propagate, reference src/lbfgs.rs:503-560) of the
synthetic diagonal quadratic of BASELINE.json config 4: n = 1e8, m = 10, More-Thuente, f64,
everything resident in HBM (device objective, no PCIe in the timed region).  It contains one
fused line-step+evaluate+g.d trial (normally exactly one), the history update, the fused
two-loop recursion and the step clamp.  With N > 1 the n-vector is sharded contiguously over
the ranks (total work fixed => "strong" scaling) and every reduction is closed by an RCCL
all-reduce of its f64 scalars.

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
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--no-prof", action="store_true", help="do not time kernels with HIP events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=4_000_000, help="sample size of the CPU baseline")
    ap.add_argument("--grid", type=int, default=0, help="workgroups per launch (0 = library default)")
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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    import rust_lbfgs_amd as R
    from rust_lbfgs_amd import _ffi, objectives

    dist = None
    torch = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from rust_lbfgs_amd.dist import sharded_context

        ctx = sharded_context(a.n, device=local_rank, kind="rccl")
    else:
        ctx = R.Context(a.n, device=local_rank)
    if a.grid:
        ctx.set_grid(a.grid)

    def barrier():
        ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    import numpy as np

    builder = R.lbfgs().with_m(a.m).with_epsilon(0.0)
    x0 = np.zeros(ctx.n_local)
    state = builder.build(x0, objectives.Quadratic(), ctx=ctx)
    restarts = 0

    def step():
        nonlocal state, restarts
        try:
            return state.propagate()
        except R.LbfgsError:
            # converged to rounding error (the line search cannot make progress): start over
            state.close()
            state = builder.build(x0, objectives.Quadratic(), ctx=ctx)
            restarts += 1
            return state.propagate()

    prefill = max(0, a.m + 2 - a.warmup)  # history must be full (bound = m) before anything is timed
    for _ in range(prefill + a.warmup):
        step()

    ncalls = 0
    if not a.no_prof:
        ctx.prof_enable(True)
        ctx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ncalls += step().ncall
    barrier()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        n_local = ctx.n_local
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None}
        if not a.no_prof:
            ns, ms_step = ctx.prof_read(_ffi.K_TWOLOOP_STEP)
            ne, ms_edge = ctx.prof_read(_ffi.K_TWOLOOP_EDGE)
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
                except Exception:
                    pass
            if nt:
                t_tl = ms_all / nt
                # 8*b passes of 8 bytes: the fused minimum that respects the dot->axpy dependency
                gbps = 64.0 * a.m * n_local / (t_tl * 1e-3) / 1e9
                roof.update(two_loop={"ms": t_tl, "algorithmic_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS,
                                      "bytes": 64 * a.m * n_local, "calls": nt})
            roof["per_iteration_ms"] = {
                "two_loop": ms_all / max(nt, 1), "history_update": ms_upd / max(a.steps, 1),
                "line_eval": ms_eval / max(a.steps, 1), "allreduce": ms_comm / max(a.steps, 1)}
        out = {
            "metric": "L-BFGS iters/sec (two-loop HBM GB/s in roofline) at n=1e8, m=10",
            "value": a.steps / dt,
            "unit": "iters/sec",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"hashed diagonal quadratic (cond 1e3), n={a.n}, m={a.m}, MoreThuente, "
                                   f"x/g/s/y sharded contiguously over {world} GPU(s)",
                       "n": a.n, "m": a.m, "n_local_rank0": n_local, "prefill_iters": prefill,
                       "line_search_trials_per_step": ncalls / max(a.steps, 1), "restarts": restarts,
                       "allreduce": "rccl" if (world > 1 or os.environ.get("LBFGS_FORCE_RCCL") == "1") else "none"},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_n, a.m, a.n)
    state.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
