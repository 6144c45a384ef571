#!/bin/bash
# round 4: split-role touching (half the waves poll, half touch) against the all-waves form, then the trace of the split form
# (history: the `nosplit` variant was -DLH_RES_TOUCH_SPLIT=0, a switch that existed while both forms were being compared (before commit f52c870); it and the all-waves form are gone; kept as the record of how profiles/r04_split_ab.log was made)
mkdir -p gpurun_out
for cfg in "1200001 6" "3000000 6" "6000000 6" "10000000 7" "12500224 10"; do
  set -- $cfg
  for v in nosplit main nosplit main; do
    if [ "$v" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v; fi
    timeout -k 10 300 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --steps 100 --repeats 5 > gpurun_out/sp.json 2> gpurun_out/sp.err || { tail -5 gpurun_out/sp.err; exit 1; }
    python - "$1" "$2" "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/sp.json").read())
r = j["roofline"]
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} {sys.argv[3]:>8}: {j['value']:8.1f} it/s  kernel {(r.get('avg_ms') or 0)*1e3:7.1f} us = {(r.get('frac') or 0)*100:4.1f} %", flush=True)
PY
  done
done
unset LBFGS_HIP_LIB_DIR
CONFIGS="12500224 10;3000000 6" bash tools/handoff_trace.sh "" "tr_main" 2>&1 | grep -E "res-trace|xcc|tr_main"
