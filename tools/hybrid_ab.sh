#!/bin/bash
# hybrid (partially on-chip) two-loop at sizes beyond the chip: builds / grids against each other, whole iterations
#   bash tools/hybrid_ab.sh "main hbm2" "216 256" [n] [m]
for v in $1; do
  for g in $2; do
    if [ "$v" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v; fi
    LBFGS_HIP_RESIDENT_GRID=$g timeout -k 10 200 python bench.py --dim ${3:-100000000} --hist ${4:-10} --no-cpu-baseline --no-vector-free --repeats 3 > gpurun_out/hy.json 2> gpurun_out/hy.err || { tail -5 gpurun_out/hy.err; exit 1; }
    python - "$v" "$g" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/hy.json").read())
r = j["roofline"]; tl = r.get("two_loop") or {}
print(f"{sys.argv[1]:>8} grid={sys.argv[2]:>4}: {j['value']:8.2f} it/s  {r.get('kernel','?')[:24]} {r.get('avg_ms',0):7.3f} ms  two-loop {tl.get('ms', 0):.3f} ms")
PY
  done
done
