#!/bin/bash
# A/B of the cross-workgroup hand-off (tagged granules vs arrival ticket) on whole iterations:  bash tools/handoff_ab.sh
for cfg in "100000 6" "1000000 6" "3000000 6" "10000000 7" "12500000 10" "100000000 10"; do
  set -- $cfg
  for h in ticket tagged; do
    LBFGS_HIP_HANDOFF=$h timeout -k 10 200 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --steps 300 --warmup 20 > gpurun_out/ho.json 2>/dev/null || exit 1
    python - "$1" "$2" "$h" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ho.json").read())
print(f"n={sys.argv[1]:>10} m={sys.argv[2]:>2} {sys.argv[3]:>7}: {j['value']:9.1f} it/s   two-loop {j['roofline']['two_loop']['ms']*1e3:8.1f} us   restarts {j['config']['restarts']}")
PY
  done
done
