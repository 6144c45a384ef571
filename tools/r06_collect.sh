#!/bin/bash
# copy the evidence of tools/r06_evidence.sh from gpurun_out/ into profiles/ under the names bench.py and the tests look for
set -e
cpy() { [ -f "$1" ] && cp "$1" "$2" && echo "  $2"; }
d=gpurun_out/prof_r06
if [ -d $d ]; then
  cpy $d/bench.json profiles/r06_bench_n1e8_m10.json; cpy $d/summary.md profiles/r06_bench_n1e8_m10.md
  cpy $d/kernel_stats.csv profiles/r06_bench_n1e8_m10_kernel_stats.csv
  cpy $d/pmc_fetch_counter_collection.csv profiles/r06_pmc_fetch_counter_collection.csv
  cpy $d/pmc_write_counter_collection.csv profiles/r06_pmc_write_counter_collection.csv
  cpy $d/pmc_traffic.json profiles/pmc_traffic.json
fi
for p in 8 4 2; do
  d=gpurun_out/prof_r06_shard_P$p
  [ -d $d ] || continue
  cpy $d/bench.json profiles/r06_shard_P${p}_bench.json; cpy $d/summary.md profiles/r06_shard_P${p}_profile.md
  cpy $d/kernel_stats.csv profiles/r06_shard_P${p}_kernel_stats.csv
  cpy $d/pmc_fetch_counter_collection.csv profiles/r06_shard_P${p}_pmc_fetch_counter_collection.csv
  cpy $d/pmc_write_counter_collection.csv profiles/r06_shard_P${p}_pmc_write_counter_collection.csv
  cpy $d/pmc_traffic.json profiles/pmc_traffic_shard_P$p.json
done
for k in 2 3 5; do
  d=gpurun_out/prof_r06_config$k
  [ -d $d ] || continue
  cpy $d/run.jsonl profiles/r06_config${k}_run.jsonl; cpy $d/summary.md profiles/r06_config${k}.md
  cpy $d/kernel_stats.csv profiles/r06_config${k}_kernel_stats.csv
  cpy $d/pmc_fetch_counter_collection.csv profiles/r06_config${k}_pmc_fetch_counter_collection.csv
  cpy $d/pmc_write_counter_collection.csv profiles/r06_config${k}_pmc_write_counter_collection.csv
  cpy $d/pmc_traffic.json profiles/pmc_traffic_config$k.json
done
# (evidence taken before summarize_profile.py wrote `resident_elements`: take it from the same run's bench line)
python3 - <<'PY'
import glob, json, os
pairs = {"profiles/pmc_traffic.json": "profiles/r06_bench_n1e8_m10.json"}
for p in (8, 4, 2):
    pairs[f"profiles/pmc_traffic_shard_P{p}.json"] = f"profiles/r06_shard_P{p}_bench.json"
for tf, bf in pairs.items():
    if not (os.path.exists(tf) and os.path.exists(bf)):
        continue
    t = json.load(open(tf))
    if t.get("resident_elements") is None:
        roof = json.loads(open(bf).read().strip().splitlines()[-1])["roofline"]
        t["resident_elements"] = roof.get("resident_elements")
        json.dump(t, open(tf, "w"), indent=1)
        print("  resident_elements ->", tf)
PY
# every summary gets a header that names the build it was taken with (the id stamped into the pmc_traffic file of the same pass)
python3 - <<'PY'
import json, os
H = {
 "profiles/r06_bench_n1e8_m10.md": ("profiles/pmc_traffic.json",
   "# rocprofv3 summary of `python bench.py` (BASELINE.json's metric: quadratic n = 1e8, m = 10, one MI355X), round 6",
   "`tools/profile_round.sh r06`: kernel statistics from a 400-step run (`r06_bench_n1e8_m10_kernel_stats.csv`), FETCH_SIZE / WRITE_SIZE from separate `--pmc` passes (raw rows: `r06_pmc_{fetch,write}_counter_collection.csv`; FETCH_SIZE x2, KiB: MI355X_MICROARCH.md, HBM section). Un-profiled line of the same command: `r06_bench_n1e8_m10.json`."),
 "profiles/r06_shard_P8_profile.md": ("profiles/pmc_traffic_shard_P8.json",
   "# rocprofv3 summary of `python bench.py --dim 12500224` -- rank 0's shard of the 8-GPU run of the metric, alone on one MI355X, round 6",
   "`DIM=12500224 tools/profile_round.sh r06_shard_P8`; raw PMC rows: `r06_shard_P8_pmc_{fetch,write}_counter_collection.csv`; `pmc_traffic_shard_P8.json` is what `bench.py` reports as `roofline.traffic` at this shard size."),
 "profiles/r06_shard_P4_profile.md": ("profiles/pmc_traffic_shard_P4.json",
   "# rocprofv3 summary of `python bench.py --dim 25000192` -- rank 0's shard of the 4-GPU run, alone on one MI355X, round 6",
   "`DIM=25000192 tools/profile_round.sh r06_shard_P4`; raw PMC rows: `r06_shard_P4_pmc_{fetch,write}_counter_collection.csv`; `pmc_traffic_shard_P4.json`."),
 "profiles/r06_shard_P2_profile.md": ("profiles/pmc_traffic_shard_P2.json",
   "# rocprofv3 summary of `python bench.py --dim 50000128` -- rank 0's shard of the 2-GPU run, alone on one MI355X, round 6",
   "`DIM=50000128 tools/profile_round.sh r06_shard_P2`; raw PMC rows: `r06_shard_P2_pmc_{fetch,write}_counter_collection.csv`; `pmc_traffic_shard_P2.json`."),
 "profiles/r06_config2.md": ("profiles/pmc_traffic_config2.json",
   "# rocprofv3 summary of BASELINE config 2 (quadratic n = 1e7, m = 7, More-Thuente; `tools/run_configs.py --only config2`), round 6",
   "`tools/profile_configs.sh r06 2`; run: `r06_config2_run.jsonl`; kernel statistics `r06_config2_kernel_stats.csv`; raw PMC rows `r06_config2_pmc_{fetch,write}_counter_collection.csv`. The two-loop is the persistent kernel with all of q on the chip (4m+1 = 29 passes)."),
 "profiles/r06_config3.md": ("profiles/pmc_traffic_config3.json",
   "# rocprofv3 summary of BASELINE config 3 (OWL-QN, L1 logistic n = 1e7, m = 6; `tools/run_configs.py --only config3`), round 6",
   "`tools/profile_configs.sh r06 3`; run: `r06_config3_run.jsonl`. The `orthantwise.rs:140-161` path: the persistent two-loop kernel's LAST step projects d onto the orthant of -pg as it writes it (25 passes: pg at the start, 2 vectors per step, d written once); `OpObjOwlLineEval<Obj, FIRST, UPD>` is one OWL-QN trial in one pass (`orthantwise.rs:70-133`: projected step, logistic evaluate -- softplus and sigmoid in hand-written f64, `ops.h` LogisticMath --, x1norm, pseudo-gradient); FIRST: it also forms the orthant of the new point (`core.rs:167-180`); UPD (round 6): it also does `IterationData::update` for its point (`lbfgs.rs:640-656`: s, y, three sums) -- neither `OpOrthantSelect` nor `OpHistUpdate` appears in an iteration: two kernels, the two-loop and the trial (4r 6w)."),
 "profiles/r06_config5.md": ("profiles/pmc_traffic_config5.json",
   "# rocprofv3 summary of BASELINE config 5 (damped L-BFGS on a Lennard-Jones system of 1e6 atoms, n = 3e6, m = 6; `tools/run_configs.py --only config5`), round 6",
   "`tools/profile_configs.sh r06 5`; run: `r06_config5_run.jsonl` (two windows: the first 35 iterations, iterations 6-305). Optimiser side: the persistent two-loop kernel (`two_loop_resident_kernel<8,...>`: 23 rounds per thread, touching depth 8), `OpHistUpdate<true>` = `IterationData::update` with the Powell-damping sums of `lbfgs.rs:664-673`, `OpDamp` = damping case 1 (`lbfgs.rs:675-680`, fires rarely on this run). The `lj_*` kernels are the user's objective kept on the device (cell list, Verlet list, evaluation): not priced against HBM. Round 5: a trial's point is formed by `lj_cells_step_check_kernel` (the list check's pass) and g.d is summed by `lj_cells_eval_kernel<.., true>`: `OpLineStep` and `OpDot` no longer appear in an iteration."),
}
for md, (tf, title, what) in H.items():
    if not (os.path.exists(md) and os.path.exists(tf)):
        continue
    body = open(md).read()
    if body.startswith("# "):
        continue  # (has its header already)
    bid = json.load(open(tf)).get("build_id")
    open(md, "w").write(f"{title}\n\nbuild `{bid}`; {what}\n\n{body}")
    print("  header ->", md)
PY
