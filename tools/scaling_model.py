#!/usr/bin/env python3
"""A strong-scaling model the 8-GPU SCALE run can be read against (profiles/rNN_scaling_model.md).

    python tools/scaling_model.py profiles/r05_bench_n1e8_m10.json P2=profiles/r05_shard_P2_bench.json \
           P4=profiles/r05_shard_P4_bench.json P8=profiles/r05_shard_P8_bench.json [--round=5] \
           [--exchange-measured-us=7.0 --exchange-measured-where="eight ranks (4 processes x 2 threads) ... (file)"]

Inputs are `bench.py` lines MEASURED ON ONE GPU: the metric's configuration (n = 1e8, m = 10) and the rank-0 shards of the
2-, 4- and 8-GPU runs alone (`bench.py --dim 50000128 / 25000192 / 12500224`: what ONE rank of such a run does between its
exchanges -- same kernels, same launch forms, no peer).  The model adds the one thing a single GPU cannot show: the
cross-rank exchanges.  Every reduction of an iteration is closed by one exchange:

    exact two-loop       2m   exchanges inside the persistent kernel's hand-offs (the first numerator comes out of the
                              history update's exchange), dependent: each is on the critical path
    history update       1    (seven sums in one message)
    line search          1 per trial (f and g.d in one message)

    t_iter(P, L) = t_iter_one_rank_alone(n/P) + N_exchanges * L            L = latency of one exchange as the kernel sees it
    (t_iter_one_rank_alone is re-priced at the trial count of the n = 1e8 run: a shard alone is a smaller problem with its own
    line searches; per-trial cost = the shard's measured line_eval ms / its trials)
    speed-up(P, L) = t_iter(1) / t_iter(P, L)

L is what `roofline.exchange_us_mean` reports in a real run (stores to P-1 mailboxes + wait for P-1 peers; it INCLUDES the
skew between ranks, which one GPU cannot produce).  The vector-free extension needs 2 exchanges per two-loop instead of 2m.
Nothing here is a measurement of more than one GPU; it says at which exchange latency the north star's 6x at 8 GPUs is lost.
"""
import json
import sys


def load(path):
    txt = open(path).read().strip().splitlines()
    for ln in reversed(txt):
        if ln.strip().startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"{path}: no JSON line")


def facts(j):
    cfg, roof = j["config"], j["roofline"]
    ext = (cfg.get("extension_vector_free_two_loop") or {})
    vf = next(iter(ext.values()), {}) if ext else {}
    pi = roof.get("per_iteration_ms") or {}
    return dict(n=cfg.get("n_local_rank0") or cfg["n"], m=cfg["m"], ips=j["value"], ms=j["ms_per_step"],
                upd_ms=pi.get("history_update"), eval_ms=pi.get("line_eval"),
                trials=cfg["line_search_trials_per_step"], two_loop_ms=(roof.get("two_loop") or {}).get("ms"),
                kernel_us=(roof.get("avg_ms") or 0) * 1e3, frac=roof.get("frac"), vf_ips=vf.get("iters_per_sec"),
                vf_two_loop_ms=vf.get("two_loop_ms"))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    measured, where, rnd = None, None, "5"
    for a in sys.argv[1:]:
        if a.startswith("--exchange-measured-us"):
            measured = float(a.split("=", 1)[1]) if "=" in a else None
        elif a.startswith("--exchange-measured-where="):
            where = a.split("=", 1)[1]
        elif a.startswith("--round="):
            rnd = a.split("=", 1)[1]
    one = facts(load(args[0]))
    shards, per_step, gated = {}, {}, {}
    for a in args[1:]:
        k, p = a.split("=", 1)
        # P<k>= the rank-0 shard of a k-GPU run alone (default launch form: the persistent kernel);
        # S<k>= the same under a 1-rank RCCL communicator with a kernel per two-loop step (LBFGS_HIP_RCCL_RESIDENT=0; or
        #       LBFGS_HIP_RESIDENT=0 without a communicator): the RCCL leg's fallback form;
        # G<k>= the same under a 1-rank RCCL communicator with the GATED exchange (the RCCL leg's default form: the persistent
        #       kernel + gate / ncclAllReduce / post on a second stream; a 1-rank all-reduce launches no kernel, so the figure
        #       contains the machinery and nothing of RCCL's own time)
        {"P": shards, "S": per_step, "G": gated}[k[0]][int(k[1:])] = facts(load(p))
    m = one["m"]
    lat = [0.0, 2.0, 5.0, 10.0, 20.0, 50.0]
    out = []
    out.append(f"# Strong-scaling model for BASELINE.json's metric (n = 1e8, m = 10) -- round {rnd}\n")
    out.append("NOT a multi-GPU measurement (no 8-GPU node has been offered to this build so far): one-GPU measurements of what a rank "
               "does between its exchanges, plus N exchanges of latency L per iteration.  Script: `tools/scaling_model.py`; inputs: the "
               "`bench.py` lines named below.\n")
    out.append("## Measured on one GPU\n")
    out.append("| what | n_local | iters/s | ms/iter | two-loop ms | dominant kernel us | kernel % of 8 TB/s | trials/iter | vector-free iters/s |")
    out.append("|---|---|---|---|---|---|---|---|---|")
    rows = [("P = 1: the metric's own configuration", one)] + [(f"one rank of P = {p}, alone", shards[p]) for p in sorted(shards)]
    for label, f in rows:
        out.append(f"| {label} | {f['n']} | {f['ips']:.1f} | {f['ms']:.3f} | {f['two_loop_ms']:.3f} | {f['kernel_us']:.1f} | "
                   f"{(f['frac'] or 0) * 100:.1f} | {f['trials']:.2f} | {f['vf_ips'] or float('nan'):.1f} |")
    out.append("")
    for form, nx_tl, key in (("exact two-loop (the product path)", 2 * m, "ms"), ("vector-free extension (opt-in)", 2, "vf")):
        out.append(f"## Predicted: {form}\n")
        out.append(f"Exchanges per iteration: {nx_tl} (two-loop) + 1 (history update) + trials (line search).\n")
        hdr = "| P | exchanges/iter | " + " | ".join(f"L = {x:g} us: iters/s (speed-up)" for x in lat) + " | L at which speed-up = 6 (P = 8) / = P/2 |"
        out.append(hdr)
        out.append("|---|---|" + "---|" * (len(lat) + 1))
        base_ms = one["ms"] if key == "ms" else (1e3 / one["vf_ips"] if one["vf_ips"] else None)
        if base_ms is None:
            out.append("| (no vector-free figure in the input lines) |")
            continue
        for p in sorted(shards):
            f = shards[p]
            t0 = f["ms"] if key == "ms" else (1e3 / f["vf_ips"] if f["vf_ips"] else None)
            if t0 is None:
                continue
            # A shard alone is another (smaller) problem: its line searches need another number of trials than the n = 1e8 run
            # whose iterations a sharded run repeats.  Re-price the line search at the metric's trial count.
            if f["eval_ms"] and f["trials"]:
                t0 += f["eval_ms"] / f["trials"] * (one["trials"] - f["trials"])
            nx = nx_tl + 1 + one["trials"]
            cells = []
            for x in lat:
                t = t0 + nx * x * 1e-3
                cells.append(f"{1e3 / t:.0f} ({base_ms / t:.1f}x)")
            target = 6.0 if p == 8 else p / 2.0
            # base/(t0 + nx*L) = target  ->  L = (base/target - t0)/nx
            lcrit = (base_ms / target - t0) / nx * 1e3
            cells.append(f"{lcrit:.0f} us (speed-up {target:g})" if lcrit > 0 else "never (below it already alone)")
            out.append(f"| {p} | {nx:.1f} | " + " | ".join(cells) + " |")
        out.append("")
    if gated:
        out.append("## Predicted: the RCCL leg, gated exchange (ncclAllReduce on a second stream under the persistent kernel)\n")
        out.append(f"Exchanges per iteration: {2 * m} (two-loop, gated) + 1 (history update) + trials (line search), the last two closed by an "
                   "all-reduce behind their kernel.  The one-rank figure already contains the gates, the posts and both kernel boundaries "
                   "of every exchange; L = what ncclAllReduce ITSELF takes for <= 32 bytes across the ranks.\n")
        lat_r = [0.0, 10.0, 15.0, 20.0, 30.0, 50.0]
        out.append("| P | one rank alone, gated machinery included: iters/s | " + " | ".join(f"L = {x:g} us: iters/s (speed-up)" for x in lat_r) +
                   " | L at which speed-up = 6 (P = 8) / = P/2 |")
        out.append("|---|---|" + "---|" * (len(lat_r) + 1))
        for p in sorted(gated):
            f = gated[p]
            t0 = f["ms"]
            if f["eval_ms"] and f["trials"]:
                t0 += f["eval_ms"] / f["trials"] * (one["trials"] - f["trials"])
            nx = 2 * m + 1 + one["trials"]
            cells = [f"{1e3 / (t0 + nx * x * 1e-3):.0f} ({one['ms'] / (t0 + nx * x * 1e-3):.1f}x)" for x in lat_r]
            target = 6.0 if p == 8 else p / 2.0
            lcrit = (one["ms"] / target - t0) / nx * 1e3
            cells.append(f"{lcrit:.0f} us (speed-up {target:g})" if lcrit > 0 else "never (below it already alone)")
            out.append(f"| {p} | {f['ips']:.0f} | " + " | ".join(cells) + " |")
        out.append("")
    if per_step:
        out.append("## Predicted: the RCCL leg's fallback form (ncclAllReduce per reduction; the two-loop as one kernel per step)\n")
        out.append(f"Exchanges per iteration: {2 * m + 1} (two-loop: every dot product is closed by an all-reduce launch of its own) + 1 "
                   "(history update) + trials (line search).  L = what one ncclAllReduce of <= 48 bytes adds to the stream "
                   "(`config.rccl.allreduce_us_mean` in a real run).\n")
        lat_r = [0.0, 10.0, 15.0, 20.0, 30.0, 50.0]
        out.append("| P | one rank alone, kernel per step: iters/s | " + " | ".join(f"L = {x:g} us: iters/s (speed-up)" for x in lat_r) +
                   " | L at which speed-up = 6 (P = 8) / = P/2 |")
        out.append("|---|---|" + "---|" * (len(lat_r) + 1))
        for p in sorted(per_step):
            f = per_step[p]
            t0 = f["ms"]
            if f["eval_ms"] and f["trials"]:
                t0 += f["eval_ms"] / f["trials"] * (one["trials"] - f["trials"])
            nx = 2 * m + 1 + 1 + one["trials"]
            cells = [f"{1e3 / (t0 + nx * x * 1e-3):.0f} ({one['ms'] / (t0 + nx * x * 1e-3):.1f}x)" for x in lat_r]
            target = 6.0 if p == 8 else p / 2.0
            lcrit = (one["ms"] / target - t0) / nx * 1e3
            cells.append(f"{lcrit:.0f} us (speed-up {target:g})" if lcrit > 0 else "never (below it already alone)")
            out.append(f"| {p} | {f['ips']:.0f} | " + " | ".join(cells) + " |")
        out.append("")
    out.append("## Reading\n")
    p8 = shards.get(8)
    if p8:
        nx = 2 * m + 1 + one["trials"]
        out.append(f"* One rank of the 8-GPU run alone does {p8['ips']:.0f} iterations/s ({one['ms'] / p8['ms']:.1f}x the single GPU): the "
                   f"shard's running vector fits on the chip (DESIGN section 3a), so the two-loop moves 4m+1 passes instead of 8m-1 over "
                   f"most of q -- strong scaling is super-linear until the exchanges are paid.")
        out.append(f"* With {nx:.1f} exchanges per iteration, every microsecond of exchange latency costs {nx * 1e-3 / p8['ms'] * 100:.1f} % "
                   f"of an iteration at P = 8.")
    if measured is not None:
        out.append(f"* Measured on ONE GPU (no xGMI in it), {where or 'several ranks sharing it'}: {measured:.1f} us per exchange inside the "
                   "persistent kernel (`roofline.exchange_us_mean`; it includes waiting for ranks that share the same HBM).  Over xGMI a "
                   "store + poll round trip is expected at 2-5 us; rank skew adds to it.")
    out.append(f"* If `SCALE_r{int(rnd):02d}.json` arrives: compare each leg's `exchange_us_mean` and `exchanges_per_two_loop` with the L column that "
               "matches, and its `two_loop_ms` with (two-loop ms of the row above + 2m * L); `config.rccl` carries the RCCL leg's figures "
               "whichever leg `value` comes from.")
    print("\n".join(out))


if __name__ == "__main__":
    main()
