#!/bin/bash
# A/B of resident-kernel builds in whole iterations:  bash tools/resident_variants_ab.sh "12500000 3000000" "main res_old" [m]
# ("main" = the in-tree library; other names = tools/bin/variants/<name>)
for n in $1; do
  for rep in 1 2; do
    for v in $2; do
      if [ "$v" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v; fi
      timeout -k 10 200 python bench.py --dim $n --hist ${3:-10} --no-cpu-baseline --no-vector-free --steps 100 --repeats 3 > gpurun_out/rv.json 2> gpurun_out/rv.err || { tail -5 gpurun_out/rv.err; exit 1; }
      python - "$n" "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/rv.json").read())
r = j["roofline"]; tl = r.get("two_loop") or {}
print(f"n={sys.argv[1]:>9} {sys.argv[2]:>10}: {j['value']:8.1f} it/s  {r.get('kernel','?')[:24]} {r.get('avg_ms',0)*1e3:7.1f} us = {r.get('achieved',0):5.0f} GB/s  two-loop {tl.get('ms', 0):.3f} ms")
PY
    done
  done
done
