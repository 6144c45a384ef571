#!/usr/bin/env python3
"""Throughput of the device-resident Lennard-Jones evaluators (the USER objective of examples/lj.rs kept in HBM):
all-pairs kernel in G pair interactions/s and as a fraction of the FP64 vector peak, and the cell-list (LJ_CELLS)
evaluation + list build at 1e6 atoms.

    python tools/lj_rates.py > gpurun_out/lj_rates.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_lbfgs_amd as R  # noqa: E402
from rust_lbfgs_amd import hotpath as H, objectives  # noqa: E402
from rust_lbfgs_amd.math import DeviceVec  # noqa: E402

# MI355X FP64 vector peak: 256 CUs x 4 SIMD x 16 lanes x 2 flop (FMA) x 2.4 GHz = 78.6 TFLOP/s (= half of the FP32
# vector figure of MI355X_MICROARCH.md, 157.3 TFLOP/s).  Per pair the kernel issues 3 sub + 3 (r^2) + 4 (Newton) +
# 2 (s6) + 2 (energy) + 4 (c) + 3 (force) = 21 FP64 vector instructions + 1 v_rcp_f64 (quarter rate: counted as 4).
FP64_PEAK_INSTR = 256 * 4 * 16 * 2.4e9  # FP64 vector instructions per second (an FMA is one instruction)
INSTR_PER_PAIR = 25


def lattice(natoms, seed=0):
    side = int(np.ceil(natoms ** (1 / 3)))
    g = np.stack(np.meshgrid(*[np.arange(side, dtype=np.float64)] * 3, indexing="ij"), -1).reshape(-1, 3)[:natoms]
    return (g * 1.12 + np.random.default_rng(seed).uniform(-0.05, 0.05, g.shape)).reshape(-1)


def timed(ctx, fn, reps):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


for natoms, reps in ((10_000, 50), (30_000, 20), (100_000, 5), (300_000, 2)):
    x = lattice(natoms)
    with R.Context(len(x)) as ctx:
        xv, gv = DeviceVec(ctx, x), DeviceVec(ctx)
        obj = objectives.LennardJones()
        dt = timed(ctx, lambda: H.objective_eval(obj, xv, gv, 0), reps)
        pairs = natoms * (natoms - 1.0)  # every pair from both ends
        print(json.dumps(dict(kernel="lj_allpairs", natoms=natoms, ms=dt * 1e3, G_pair_interactions_per_s=pairs / dt / 1e9,
                              fp64_vector_instr_per_pair=INSTR_PER_PAIR,
                              frac_of_fp64_vector_peak=pairs * INSTR_PER_PAIR / dt / FP64_PEAK_INSTR)), flush=True)
        xv.free(); gv.free()

x = lattice(1_000_000)
with R.Context(len(x)) as ctx:
    xv, gv = DeviceVec(ctx, x), DeviceVec(ctx)
    obj = objectives.LennardJonesCells(2.5, 0.3)
    dt = timed(ctx, lambda: H.objective_eval(obj, xv, gv, 0), 20)
    _, _, longest = ctx.lj_cells_stats()
    # a rebuild per call: alternate between two configurations that are more than skin/2 apart
    x2 = x + 0.2
    xa, xb = DeviceVec(ctx, x), DeviceVec(ctx, x2)
    flip = [0]

    def rebuild_eval():
        flip[0] ^= 1
        H.objective_eval(obj, xb if flip[0] else xa, gv, 0)

    dt2 = timed(ctx, rebuild_eval, 10)
    print(json.dumps(dict(kernel="lj_cells", natoms=1_000_000, cutoff=2.5, skin=0.3, longest_list=longest,
                          eval_ms_list_valid=dt * 1e3, stale_check_plus_rebuild_plus_eval_ms=dt2 * 1e3,
                          # (the staleness check runs before the evaluation: a stale list costs the check, not an evaluation)
                          rebuild_ms=(dt2 - dt) * 1e3)), flush=True)
    for v in (xv, gv, xa, xb):
        v.free()
