#!/bin/bash
# The tagged granules in uncached memory (default) against plain hipMalloc (LBFGS_HIP_GRAN_CACHED=1), whole iterations:
#   bash tools/gran_ab.sh            (sizes: config 5's optimiser side, config 2, config 3, the 8-GPU run's shard, 1e5, 1e8)
mkdir -p gpurun_out
for cfg in "3000000 6" "10000000 7" "10000000 6" "12500000 10" "100000 6" "1000000 6" "100000000 10"; do
  set -- $cfg
  for rep in 1 2; do
    for cached in 1 0; do
      if [ $cached = 1 ]; then export LBFGS_HIP_GRAN_CACHED=1; else unset LBFGS_HIP_GRAN_CACHED; fi
      timeout -k 10 200 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --steps 100 --repeats 3 > gpurun_out/ga.json 2> gpurun_out/ga.err || { tail -5 gpurun_out/ga.err; exit 1; }
      python - "$1" "$2" "$cached" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ga.json").read())
r = j["roofline"]; tl = r.get("two_loop") or {}
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} granules {'cached  ' if sys.argv[3]=='1' else 'uncached'}: {j['value']:8.1f} it/s  {r.get('kernel','?')[:24]} {(r.get('avg_ms') or 0)*1e3:7.1f} us = {r.get('achieved') or 0:5.0f} GB/s ({(r.get('frac') or 0)*100:4.1f} %)  two-loop {tl.get('ms', 0):.3f} ms", flush=True)
PY
    done
  done
done
