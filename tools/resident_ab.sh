#!/bin/bash
# the on-chip-resident two-loop kernel (resident.h) vs the launch-per-step path, whole iterations:
#   bash tools/resident_ab.sh "1200000 3000000 8000000" 10
for n in $1; do
  for r in 0 1 0 1; do
    LBFGS_HIP_RESIDENT=$r timeout -k 10 200 python bench.py --dim $n --hist ${2:-10} --no-cpu-baseline --no-vector-free --steps 60 --repeats 3 > gpurun_out/rs.json 2> gpurun_out/rs.err || { tail -5 gpurun_out/rs.err; exit 1; }
    python - "$n" "$r" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/rs.json").read())
tl = j["roofline"].get("two_loop") or {}
print(f"n={sys.argv[1]:>9} resident={sys.argv[2]}: {j['value']:8.1f} it/s  two-loop {tl.get('ms', 0):.3f} ms  ({tl.get('algorithmic_GBps', 0):.0f} GB/s algorithmic)  restarts {j['config']['restarts']}")
PY
  done
done
