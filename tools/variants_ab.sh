#!/bin/bash
# In-situ A/B of library variants built by tools/build_variants.sh:  bash tools/variants_ab.sh "m1u1 m1u2 ..." "208 216 224"
for v in $1; do
  for g in $2; do
    env LBFGS_HIP_LIB_DIR=tools/bin/variants/$v LBFGS_HIP_GRID_K0=$g timeout -k 10 150 python bench.py --no-cpu-baseline --no-vector-free --steps 40 > gpurun_out/va.json 2>/dev/null || exit 1
    python - "$v" "$g" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/va.json").read())
r = j["roofline"]
print(f"{sys.argv[1]} grid {sys.argv[2]:>4}: {j['value']:6.2f} it/s  step kernel {r['avg_ms']*1e3:6.1f} us = {r['achieved']:.0f} GB/s  two-loop {r['two_loop']['ms']:.3f} ms")
PY
  done
done
