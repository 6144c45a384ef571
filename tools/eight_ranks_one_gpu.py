#!/usr/bin/env python3
"""World = 8 -- the metric's own world -- rehearsed on ONE GPU.

The pool admits six GPU processes per card, so `bench.py --gpus 8 --device 0` (eight rank processes) cannot run there.  Here
the eight ranks are P processes x T host threads (default 4 x 2): every rank has its own context, stream, shard and mailbox,
exactly as one process per GPU has; ranks that share a process reach each other's device-placed mailbox through the pointer
(the library recognises handles of its own process: csrc/context.hip, local_mbox_lookup), ranks in other processes through
HIP IPC, host-placed mailboxes through shared memory either way.  What runs is bench.py's own rank code -- make_context()
with its start-up self-tests and measure() -- behind a rendezvous made of files (barriers, gathers: FileGroup below) instead of
torch.distributed.  What this cannot rehearse is the supervisor's eight child processes (tests/test_bench_record_cpu.py runs
that launch form, world 8, on the CPU test double), RCCL (one communicator per GPU) and anything that crosses xGMI.

    python tools/eight_ranks_one_gpu.py [--dim 100000000] [--procs 4] [--threads 2] [--legs p2p,p2p-host,callback]
        1. whole optimisations of eight ranks against the single-rank ORACLE (an on-chip shard size and a hybrid one, quadratic
           and OWL-QN), per leg;
        2. the bench leg at --dim (n = 1e8: seven shards of 12 500 224 elements and the short last one of 12 498 432), per leg:
           ranks_seen, mailboxes mapped, exchanges per two-loop, microseconds per exchange, iterations/sec;
    -> gpurun_out/bench_eight_ranks_sharing_one_gpu.json (one record: the trajectory checks, every leg's bench line).
LBFGS_HIP_RESIDENT_GRID = 24 workgroups per rank (8 x 24 = 192 of 256 CUs): every rank "owns its GPU" as far as the
persistent kernel with the exchange inside its hand-offs is concerned.
"""
import argparse
import json
import os
import pickle
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [  # whole runs against the oracle: 24 workgroups keep 1.18e6 elements on the chip
    dict(name="q8_on_chip", n=8 * 1_000_000 + 77, m=6, iters=12, objective="quadratic"),
    dict(name="q8_hybrid", n=8 * 2_600_000 + 5, m=5, iters=10, objective="quadratic"),
    dict(name="owl8", n=8 * 400_000 + 1, m=6, iters=12, objective="logistic", owl=[0.5, 300_000, 2_900_000]),
]


if os.environ.get("LBFGS_EIGHT_SMALL") == "1":  # the CPU suite's dry run of this tool on the test double: the same cases, small
    CASES = [dict(c, n=c["n"] // 100 + 3, **({"owl": [0.5, 3_000, 29_000]} if c.get("owl") else {})) for c in CASES]


class FileGroup:
    """rank / world / all_gather_object / broadcast_object / barrier over a directory: works between threads and processes
    alike.  Every rank calls the collectives in the same order (a sequence number names each one)."""

    def __init__(self, path, rank, world):
        self.path, self.rank, self.world, self.seq = path, rank, world, 0

    def all_gather_object(self, obj, timeout=600.0):
        self.seq += 1
        mine = os.path.join(self.path, f"c{self.seq}_r{self.rank}")
        with open(mine + ".tmp", "wb") as f:
            pickle.dump(obj, f)
        os.rename(mine + ".tmp", mine)
        out, t0 = [None] * self.world, time.monotonic()
        for r in range(self.world):
            p = os.path.join(self.path, f"c{self.seq}_r{r}")
            while not os.path.exists(p):
                if time.monotonic() - t0 > timeout:
                    raise TimeoutError(f"rank {self.rank}: collective {self.seq}: rank {r} did not arrive")
                time.sleep(0.0002)
            with open(p, "rb") as f:
                out[r] = pickle.load(f)
        old = os.path.join(self.path, f"c{self.seq - 2}_r{self.rank}")  # everybody has read it: two collectives ago
        if os.path.exists(old):
            os.unlink(old)
        return out

    def broadcast_object(self, obj, src=0):
        return self.all_gather_object(obj if self.rank == src else None)[src]

    def barrier(self):
        self.all_gather_object(None)


class RankEnv:
    """what bench.make_context / bench.measure expect of bench.Env"""

    def __init__(self, a, group, dev):
        self.a, self.group, self.world, self.rank, self.local_rank, self.dev = a, group, group.world, group.rank, group.rank, dev
        self.dist = self.torch = None
        self.backend = "files"

    def barrier(self, ctx=None):
        if ctx is not None:
            try:
                ctx.sync()
            except Exception:  # noqa: BLE001
                pass
        self.group.barrier()

    def reduce(self, v, op):
        vals = self.group.all_gather_object(v)
        return {"MAX": max, "MIN": min, "SUM": sum}[op](vals)

    def finish(self):
        self.group.barrier()


def trajectory(env, case, kind):
    """one whole optimisation of this rank's shard; rank 0 returns the comparison with the oracle's rows"""
    import numpy as np

    import rust_lbfgs_amd as R
    from rust_lbfgs_amd import dist as D, objectives

    g = env.group
    ctx = D.sharded_context(case["n"], device=env.dev, kind={"p2p": "p2p-device"}.get(kind, kind), process_group=g,
                            exclusive_device=True)
    lo, hi = D.shard_range(case["n"], g.rank, g.world)
    b = R.lbfgs().with_m(case["m"]).with_max_iterations(case["iters"]).with_epsilon(0.0)
    if case.get("owl"):
        b = b.with_orthantwise(*case["owl"])
    ev = objectives.Quadratic() if case["objective"] == "quadratic" else objectives.Logistic()
    rows, err = [], ""
    try:
        st = b.build(np.zeros(hi - lo), ev, ctx=ctx)
        # (ranks that share a process: every allocation is over before any kernel waits for a peer -- hipMalloc / hipFree of one
        # thread can wait for every stream of the process, the sibling rank's included)
        g.barrier()
        try:
            while not st.is_converged():
                p = st.propagate()
                rows.append([p.niter, p.neval, p.ncall, p.fx, p.xnorm, p.gnorm, p.step])
        except R.LbfgsError as e:
            err = str(e)
        xs = st.download("x")
        g.barrier()  # (... and nobody frees while a sibling's kernels may still be waiting for this rank)
        st.close()
    except R.LbfgsError as e:
        err, xs = str(e), np.zeros(hi - lo)
    ci = ctx.comm_info()
    mine = dict(rows=rows, err=err, resident=ctx.resident_two_loops(), on_chip=ctx.resident_elements(), n_local=hi - lo,
                ranks_seen=ci["ranks_seen"], peers=(ci["peers_device"], ci["peers_host"]))
    ctx.close()
    alls = g.all_gather_object(mine)
    xall = g.all_gather_object(xs)
    if g.rank != 0:
        return None
    ref = json.load(open(os.path.join(g.path, f"oracle_{case['name']}.json")))
    x = np.concatenate(xall)
    ok = all(o["err"] == "" and o["rows"] == alls[0]["rows"] for o in alls) and len(rows) == len(ref["rows"])
    worst = 0.0
    for got, want in zip(rows, ref["rows"]):
        ok = ok and got[:3] == want[:3]
        worst = max(worst, max(abs(u - v) / max(abs(v), 1e-6) for u, v in zip(got[3:], want[3:])))
    xref = np.load(os.path.join(g.path, f"oracle_{case['name']}_x.npy"))
    xerr = float(np.max(np.abs(x - xref)) / max(np.max(np.abs(xref)), 1e-12))
    # (the callback communicator is the host's: the library sees no peer itself and reports 0)
    seen_ok = all(o["ranks_seen"] == g.world for o in alls) if kind.startswith("p2p") else True
    ok = bool(ok and worst <= 1e-9 and xerr <= 1e-9 and seen_ok)
    return dict(case=case["name"], n=case["n"], leg=kind, ok=ok, iterations=len(rows), worst_scalar_deviation=worst, x_deviation=xerr,
                identical_rows_on_every_rank=all(o["rows"] == alls[0]["rows"] for o in alls),
                resident_launches=[o["resident"] for o in alls], on_chip_elements=[o["on_chip"] for o in alls],
                shard_elements=[o["n_local"] for o in alls], ranks_seen=[o["ranks_seen"] for o in alls],
                peers_device_host=[o["peers"] for o in alls], errors=[o["err"][:80] for o in alls if o["err"]])


def rank_main(a, rank, world, path, out):
    import bench

    try:
        group = FileGroup(path, rank, world)
        env = RankEnv(a, group, a.device)
        res = {"trajectories": [], "legs": {}}
        for leg in a.legs:
            if not a.no_trajectories:
                for case in CASES:
                    r = trajectory(env, case, leg)
                    if rank == 0:
                        res["trajectories"].append(r)
                        print(f"[eight] {leg:9s} {case['name']:11s} {'ok' if r['ok'] else 'FAILED'}: worst scalar {r['worst_scalar_deviation']:.2e}, "
                              f"x {r['x_deviation']:.2e}, on chip {r['on_chip_elements'][0]} of {r['shard_elements'][0]}, ranks seen {r['ranks_seen']}",
                              file=sys.stderr, flush=True)
            t0 = time.monotonic()
            ctx, label = bench.make_context(env, leg)
            line = None
            if ctx is not None:
                r = bench.measure(env, ctx, label, repeats=a.repeats)
                if r is not None and rank == 0:
                    if getattr(ctx, "p2p_placement", None):
                        r["mailboxes"] = ctx.p2p_placement
                    line = bench.compose(a, world, [r], {})
                    line["config"]["p2p_mailboxes"] = r.get("mailboxes")
                ctx.close()
            if rank == 0:
                res["legs"][leg] = {"seconds": round(time.monotonic() - t0, 1), "line": line}
                if line:
                    ci = line["config"]["comm_info"]
                    print(f"[eight] {leg:9s} bench n={a.n}: {line['value']:.1f} it/s, ranks_seen {ci['ranks_seen']}, mailboxes "
                          f"{ci['peers_device']}+{ci['peers_host']}, {ci['exchanges_per_two_loop']} exchanges per two-loop of "
                          f"{ci['exchange_us_mean']} us, two-loop {line['roofline'].get('two_loop', {}).get('ms')} ms", file=sys.stderr, flush=True)
                else:
                    print(f"[eight] {leg:9s} bench: no result", file=sys.stderr, flush=True)
        env.finish()
        out[rank] = res
    except Exception as e:  # noqa: BLE001
        import traceback

        traceback.print_exc()
        out[rank] = {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", dest="n", type=int, default=100_000_000)
    ap.add_argument("--hist", dest="m", type=int, default=10)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--legs", default="p2p,p2p-host,callback")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--resident-grid", type=int, default=24)
    ap.add_argument("--no-trajectories", action="store_true")
    ap.add_argument("--_proc", dest="proc", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--_dir", dest="dir", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    a.legs = a.legs.split(",")
    # what bench.measure / compose read besides the above
    a.gpus, a.no_prof, a.prof_every, a.line_eval, a.exclusive_device, a.min_timed_seconds = a.procs * a.threads, False, 5, 2, 1, 5.0
    world = a.procs * a.threads
    if a.proc >= 0:  # one of the P processes: T rank threads
        import rust_lbfgs_amd  # noqa: F401
        from rust_lbfgs_amd import _ffi

        if os.environ.get("LBFGS_TEST_BACKEND") == "mock":  # dry-run of this tool's logic on the CPU test double (callback leg only)
            from tests.support import mock

            mock.install()
        else:
            _ffi.load()  # (once, before the threads)
        out = {}
        ths = [threading.Thread(target=rank_main, args=(a, a.proc * a.threads + t, world, a.dir, out)) for t in range(a.threads)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if a.proc == 0:
            json.dump(out.get(0), open(os.path.join(a.dir, "result.json"), "w"))
        return 0 if all("error" not in (v or {"error": 1}) for v in out.values()) and len(out) == a.threads else 1

    # ---- the parent: oracle trajectories on the CPU (no GPU here), then the P processes
    import numpy as np

    from tests.test_distributed_cpu import oracle_rows

    t_start = time.monotonic()
    with tempfile.TemporaryDirectory(prefix="eight_ranks_") as d:
        if not a.no_trajectories:
            for case in CASES:
                rows, x = oracle_rows(case)
                json.dump({"rows": rows}, open(os.path.join(d, f"oracle_{case['name']}.json"), "w"))
                np.save(os.path.join(d, f"oracle_{case['name']}_x.npy"), x)
        env = dict(os.environ, LBFGS_HIP_RESIDENT_GRID=str(a.resident_grid), OMP_NUM_THREADS="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--_dir", d]
        procs = [subprocess.Popen(cmd + ["--_proc", str(i)], env=env, cwd=ROOT) for i in range(a.procs)]
        rcs = []
        for p in procs:
            try:
                rcs.append(p.wait(timeout=900))
            except subprocess.TimeoutExpired:
                p.kill()
                rcs.append(-9)
        res = json.load(open(os.path.join(d, "result.json"))) if os.path.exists(os.path.join(d, "result.json")) else None
    ok = res is not None and "error" not in res and all(rc == 0 for rc in rcs) and all(t["ok"] for t in res["trajectories"]) and \
        all(v["line"] is not None for v in res["legs"].values())
    rec = {"what": f"{world} ranks = {a.procs} processes x {a.threads} threads sharing ONE GPU (device {a.device}), "
                   f"{a.resident_grid} workgroups of the persistent kernel each; bench.py's make_context + measure per leg",
           "ok": bool(ok), "exit_codes": rcs, "seconds": round(time.monotonic() - t_start, 1), "result": res}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "bench_eight_ranks_sharing_one_gpu.json"), "w"), indent=1)
    print(json.dumps({"ok": rec["ok"], "seconds": rec["seconds"], "exit_codes": rcs,
                      "legs": {k: (v["line"]["value"] if v["line"] else None) for k, v in (res or {"legs": {}})["legs"].items()}}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
