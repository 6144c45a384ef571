#!/bin/bash
# In-situ grid sweep for ONE kernel class of bench.py (classes: include/lbfgs_hip.h LBFGS_HIP_K_*; 2 = update, 4 = objective eval/probe):
#   bash tools/grid_class_ab.sh 2 "216 256 384 512 768"
k=$1
for g in $2; do
  env LBFGS_HIP_GRID_K$k=$g timeout -k 10 150 python bench.py --no-cpu-baseline --no-vector-free --steps 40 --prof-every 1 > gpurun_out/gk.json 2>/dev/null || exit 1
  python - "$k" "$g" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/gk.json").read())
p = j["roofline"]["per_iteration_ms"]
print(f"class {sys.argv[1]} grid {sys.argv[2]:>5}: {j['value']:6.2f} it/s  update {p['history_update']:.3f} ms  trials {p['line_eval']:.3f} ms  two-loop {p['two_loop']:.3f} ms")
PY
done
