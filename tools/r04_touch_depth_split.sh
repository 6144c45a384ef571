#!/bin/bash
# round 4: touch depth (LBFGS_HIP_RESIDENT_TOUCH) re-swept with the split-role hand-off -- the all-waves form's optimum (8 rounds
# below 64 rounds per thread, 16 from there) was set by touches delaying polls, which no longer happens
mkdir -p gpurun_out
for cfg in "1200001 6" "2000000 6" "3000000 6" "4500000 6" "6000000 6" "8000000 6" "10000000 7" "12500224 10"; do
  set -- $cfg
  for t in 0 4 8 12 16 8 16; do
    LBFGS_HIP_RESIDENT_TOUCH=$t timeout -k 10 300 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --steps 100 --repeats 5 > gpurun_out/td.json 2> gpurun_out/td.err || { tail -5 gpurun_out/td.err; exit 1; }
    python - "$1" "$2" "$t" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/td.json").read())
r = j["roofline"]
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} touch={sys.argv[3]:>2}: {j['value']:8.1f} it/s  kernel {(r.get('avg_ms') or 0)*1e3:7.1f} us = {(r.get('frac') or 0)*100:4.1f} %", flush=True)
PY
  done
done
