#!/bin/bash
# copy the evidence of tools/r04_evidence.sh from gpurun_out/ into profiles/ under the names bench.py and the tests look for
set -e
cpy() { [ -f "$1" ] && cp "$1" "$2" && echo "  $2"; }
d=gpurun_out/prof_r04
if [ -d $d ]; then
  cpy $d/bench.json profiles/r04_bench_n1e8_m10.json; cpy $d/summary.md profiles/r04_bench_n1e8_m10.md
  cpy $d/kernel_stats.csv profiles/r04_bench_n1e8_m10_kernel_stats.csv
  cpy $d/pmc_fetch_counter_collection.csv profiles/r04_pmc_fetch_counter_collection.csv
  cpy $d/pmc_write_counter_collection.csv profiles/r04_pmc_write_counter_collection.csv
  cpy $d/pmc_traffic.json profiles/pmc_traffic.json
fi
for p in 8 4 2; do
  d=gpurun_out/prof_r04_shard_P$p
  [ -d $d ] || continue
  cpy $d/bench.json profiles/r04_shard_P${p}_bench.json; cpy $d/summary.md profiles/r04_shard_P${p}_profile.md
  cpy $d/kernel_stats.csv profiles/r04_shard_P${p}_kernel_stats.csv
  cpy $d/pmc_fetch_counter_collection.csv profiles/r04_shard_P${p}_pmc_fetch_counter_collection.csv
  cpy $d/pmc_write_counter_collection.csv profiles/r04_shard_P${p}_pmc_write_counter_collection.csv
  cpy $d/pmc_traffic.json profiles/pmc_traffic_shard_P$p.json
done
for k in 2 3 5; do
  d=gpurun_out/prof_r04_config$k
  [ -d $d ] || continue
  cpy $d/run.jsonl profiles/r04_config${k}_run.jsonl; cpy $d/summary.md profiles/r04_config${k}.md
  cpy $d/kernel_stats.csv profiles/r04_config${k}_kernel_stats.csv
  cpy $d/pmc_fetch_counter_collection.csv profiles/r04_config${k}_pmc_fetch_counter_collection.csv
  cpy $d/pmc_write_counter_collection.csv profiles/r04_config${k}_pmc_write_counter_collection.csv
  cpy $d/pmc_traffic.json profiles/pmc_traffic_config$k.json
done
# (evidence taken before summarize_profile.py wrote `resident_elements`: take it from the same run's bench line)
python3 - <<'PY'
import glob, json, os
pairs = {"profiles/pmc_traffic.json": "profiles/r04_bench_n1e8_m10.json"}
for p in (8, 4, 2):
    pairs[f"profiles/pmc_traffic_shard_P{p}.json"] = f"profiles/r04_shard_P{p}_bench.json"
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
