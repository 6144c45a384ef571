#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of `bench.py` into profiles/ (kernel stats + PMC HBM traffic).

    python tools/summarize_profile.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <n_local> <out.md> [tag] [m] [bench.json]
(bench.json: the un-profiled line of the same command; its roofline.bytes_per_launch prices the resident kernel, whose
on-chip share of q depends on the shard size -- without it (tools/profile_configs.sh: other commands than bench.py) the
resident kernel is priced as fully on-chip: 4m+1 passes, under OWL-QN too -- the fused write-out streams pg with the last step)

Every pmc_traffic.json it writes names the build the passes were made with (LBFGS_HIP_BUILD_ID in the environment, else
lbfgs_hip_build_id() of the in-tree library): bench.py reports roofline.traffic_build_id / traffic_is_current from it and
tests/test_bench_record_cpu.py requires the committed files to belong to the checked-out sources.

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are collected in SEPARATE passes (TCC slots), both are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.
"""
import csv
import glob
import re
import sys
from collections import defaultdict

# algorithmic passes (reads, writes) of one n-vector per launch, per kernel shape
M = 10  # history length of the profiled run (the resident kernel's passes depend on it)
PASSES = [
    (r"two_loop_resident_kernel<", None, "the whole two-loop as one kernel, q on chip: g, every s and y twice (less one), d written"),
    (r"OpTwoLoopStep<(true|false), false, 0>", (3, 1), "two-loop step  q+=c*u; out=v.q"),
    (r"OpTwoLoopStep<(true|false), true, 1>", (2, 1), "two-loop gamma transition"),
    (r"OpTwoLoopStep<false, false, 2>", (3, 1), "two-loop last step + ||d||^2 + g.d (g re-read for the next dginit)"),
    (r"OpTwoLoopFirst", (2, 0), "two-loop first dot s.(-g)"),
    (r"OpHistUpdateFromStep<", (3, 4), "accepted step: x,g + history update s,y + 7 sums"),
    (r"OpObjLineProbe<", (2, 0), "trial step: f and g.d, nothing written"),
    (r"OpHistUpdate<", (4, 2), "history update s,y + 5 sums"),
    (r"OpObjLineEval<", (2, 2), "line step + quadratic eval + g.d"),
    (r"OpObjEval<", (1, 1), "objective eval"),
    (r"OpDot[,>]", (2, 0), "dot (dginit)"),
    (r"OpNorms2", (2, 0), "norms"),
    (r"OpCopy<", (1, 1), "copy / ncopy"),
    # OWL-QN (config 3)
    (r"OpObjOwlLineEval<.*, true, true>", (4, 6), "FIRST OWL-QN trial of a search that also does IterationData::update for its point (lbfgs.rs:640-656: s, y, s.s, y.s, y.y): orthant + line step + projection + eval + x1norm + pseudo-gradient + g.d + update"),
    (r"OpObjOwlLineEval<.*, false, true>", (4, 5), "OWL-QN trial that also does IterationData::update for its point (lbfgs.rs:640-656)"),
    (r"OpObjOwlLineEval<.*, true, false>", (3, 4), "FIRST OWL-QN trial of a search: the orthant of the new point (core.rs:167-180) + line step + projection + eval + x1norm + pseudo-gradient + g.d"),
    (r"OpObjOwlLineEval<.*, false, false>", (3, 3), "OWL-QN trial: line step + projection + eval + x1norm + pseudo-gradient + g.d (orthantwise.rs:70-133)"),
    (r"OpObjOwlLineEval<[^,]*, true>", (3, 4), "FIRST OWL-QN trial of a search (round-5 builds)"),
    (r"OpObjOwlLineEval<", (3, 3), "OWL-QN trial (round-5 builds)"),
    (r"OpOrthantSelect", (2, 1), "orthant of the new point (core.rs:167-180)"),
    (r"OpOwlPost", (2, 1), "x1norm + pseudo-gradient (orthantwise.rs:70-112)"),
    (r"OpConstrainDir", (2, 1), "constrain_search_direction (orthantwise.rs:140-161)"),
    (r"OpLineStep<", (2, 1), "take_line_step (core.rs:155-164)"),
    # damping (config 5)
    (r"OpDamp", (2, 1), "Powell damping case 1: y = (1-theta) bs + theta y (lbfgs.rs:675-680)"),
    (r"OpAxpy[,>]", (2, 1), "axpy"),
    (r"OpScale[,>]", (1, 1), "scale"),
    (r"OpDiff", (2, 1), "diff"),
    (r"OpNrm2", (1, 0), "squared norm"),
    (r"OpFill", (0, 1), "fill"),
]


def shape(name):
    for pat, rw, label in PASSES:
        if re.search(pat, name):
            return (rw if rw is not None else (4 * M, 1)), label
    return None, None


def short(name):
    m = re.search(r"stream_kernel<lh::(.*?)(, \d+, \d+u, \d+u, \d+, \d+(, (true|false))?)?>\(", name)
    if m:
        return "stream_kernel<" + m.group(1) + ">"
    m = re.search(r"(two_loop_resident_kernel<\d+, (true|false)(, (true|false))?>)", name)
    if m:
        return m.group(1)
    m = re.search(r"lh::(lj_\w+)", name)
    return m.group(1) if m else name[:60]


def full_launches(values):
    """The resident kernel's work depends on the recursion's depth: the first m iterations of a run (history not yet
    full) launch it with fewer steps.  Only the FULL launches are the kernel the roofline prices: those within 3 % of
    the largest value.  (Every other kernel does the same work on every launch.)"""
    top = max(values)
    return [v for v in values if v >= 0.97 * top]


def pmc(dirname, counter):
    agg = defaultdict(list)
    for f in glob.glob(dirname + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        if "two_loop_resident_kernel" in k:
            v = full_launches(v)
        out[k] = sum(v) / len(v)
    return out


def trace_durations(stats_dir):
    """per-dispatch durations (us) from the kernel trace, by short kernel name"""
    d = defaultdict(list)
    for f in glob.glob(stats_dir + "/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d


def main():
    stats_dir, fdir, wdir, n_local, out = sys.argv[1:6]
    tag = sys.argv[6] if len(sys.argv) > 6 else "rXX"
    global M
    M = int(sys.argv[7]) if len(sys.argv) > 7 else 10
    resident_bytes = resident_elements = None
    if len(sys.argv) > 8:
        try:
            import json as _json
            roof = _json.loads(open(sys.argv[8]).read().strip().splitlines()[-1])["roofline"]
            if "two_loop_resident_kernel" in roof.get("kernel", ""):
                resident_bytes = roof["bytes_per_launch"]
                resident_elements = roof.get("resident_elements")
        except Exception:  # noqa: BLE001
            pass
    n_local = int(n_local)
    fetch = pmc(fdir, "FETCH_SIZE")
    write = pmc(wdir, "WRITE_SIZE")
    rows = []
    for f in glob.glob(stats_dir + "/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    lines = ["| kernel | calls | avg us | algorithmic passes (r+w) | algorithmic GB/launch | GB/s | % of 8 TB/s | "
             "PMC read GB (FETCH_SIZE x2) | PMC write GB (WRITE_SIZE) | PMC/algorithmic |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    durs = trace_durations(stats_dir)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        nm = short(r["Name"])
        rw, label = shape(r["Name"])
        avg_us = float(r["AverageNs"]) / 1e3
        if "two_loop_resident_kernel" in nm and durs.get(nm):
            # within 5 % of the median = the launches with the full recursion depth (see full_launches); the stats file's plain
            # average (kept in kernel_stats.csv) also counts the shallower launches of the first m iterations
            ref = sorted(durs[nm])[len(durs[nm]) // 2]  # (most launches of the profiled run are full-depth: the median is one)
            full = [v for v in durs[nm] if abs(v - ref) <= 0.05 * ref]
            label += f"; avg of the {len(full)} full-depth launches of {len(durs[nm])} (all launches: {avg_us:.1f} us)"
            avg_us = sum(full) / len(full)
        if rw is None:
            lines.append(f"| `{nm}` | {r['Calls']} | {avg_us:.1f} | - | - | - | - | - | - | - |")
            continue
        gb = (rw[0] + rw[1]) * 8 * n_local / 1e9
        rwtxt = f"{rw[0]}r+{rw[1]}w"
        if "two_loop_resident_kernel" in nm and resident_bytes:
            gb = resident_bytes / 1e9
            rwtxt = "(see left)"
            label += f"; {gb:.2f} GB = 4m+1 passes over the on-chip part of q, 8m-1 over the rest"
        elif "two_loop_resident_kernel" in nm:
            gb = (4 * M + 1) * 8 * n_local / 1e9
            rwtxt = f"{4 * M}r+1w"
            label += f"; {gb:.2f} GB = 4m+1 passes (all of q on the chip)"
        gbps = gb / (avg_us * 1e-6)
        fr = fetch.get(nm)
        wr = write.get(nm)
        fr_gb = fr * 1024 * 2 / 1e9 if fr is not None else None
        wr_gb = wr * 1024 / 1e9 if wr is not None else None
        ratio = (fr_gb + wr_gb) / gb if fr_gb is not None and wr_gb is not None else None
        lines.append(f"| `{nm}` ({label}) | {r['Calls']} | {avg_us:.1f} | {rwtxt} | {gb:.2f} | {gbps:.0f} | "
                     f"{gbps / 80:.1f} | {fr_gb:.3f} | {wr_gb:.3f} | {ratio:.3f} |" if ratio is not None else
                     f"| `{nm}` ({label}) | {r['Calls']} | {avg_us:.1f} | {rwtxt} | {gb:.2f} | {gbps:.0f} | "
                     f"{gbps / 80:.1f} | - | - | - |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # the dominant kernel's measured HBM bytes per launch, for bench.py's roofline.traffic
    import json
    import os

    dom = [k for k in fetch if re.search(r"two_loop_resident_kernel<", k)]
    kname = "two_loop_resident_kernel<ER,NT>"
    if not dom:
        dom = [k for k in fetch if re.search(r"OpTwoLoopStep<false, false, 0", k)]
        kname = "stream_kernel<OpTwoLoopStep<*,false,0>>"
    if dom and dom[0] in write:
        rd, wr = fetch[dom[0]] * 1024 * 2, write[dom[0]] * 1024
        build_id = os.environ.get("LBFGS_HIP_BUILD_ID")
        if not build_id:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
            import rust_lbfgs_amd  # noqa: F401
            from rust_lbfgs_amd import _build
            build_id = _build.embedded_id(_build.HIP_LIB, "LBFGS_HIP_BUILD_ID")
        full_us = None
        for k, v in durs.items():
            if k == dom[0] and v:
                ref = sorted(v)[len(v) // 2]
                fl = [x for x in v if abs(x - ref) <= 0.05 * ref]
                full_us = sum(fl) / len(fl)
        algo = resident_bytes if (resident_bytes and "resident" in kname) else (
            (4 * M + 1) * 8 * n_local if "resident" in kname else 32 * n_local)
        # which command the passes profiled: bench.py's roofline.traffic only accepts its own (tools/profile_round.sh sets it;
        # tools/profile_configs.sh profiles tools/run_configs.py -- other objectives, OWL-QN, damping -- at sizes bench.py can
        # also be asked to run)
        command = os.environ.get("LBFGS_PROFILE_COMMAND") or ("bench.py" if len(sys.argv) > 8 else "unknown")
        json.dump({"build_id": build_id, "command": command, "resident_elements": resident_elements, "algorithmic_bytes_per_launch": round(algo),
                   "rocprof_avg_us_full_depth_launches": full_us,"_source": f"profiles/{tag}_pmc_fetch_counter_collection.csv + profiles/{tag}_pmc_write_counter_collection.csv "
                              "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python3 bench.py`, averaged over the "
                              "dispatches of the kernel; FETCH_SIZE x2: gfx950 correction of MI355X_MICROARCH.md section HBM; KiB)",
                   "kernel": kname, "n_local": n_local, "m": M,
                   "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                   "traffic_bytes_per_launch": round(rd + wr)},
                  open(os.path.join(os.path.dirname(out), "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
