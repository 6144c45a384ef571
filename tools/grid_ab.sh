#!/bin/bash
# A/B of the launch grid for every kernel class (per_iteration_ms of bench.py): bash tools/grid_ab.sh "216 256 432 512 1024"
for g in $1; do
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-vector-free --steps 20 --grid $g > gpurun_out/grid_$g.json 2> gpurun_out/grid_$g.err || exit 1
  python - "$g" <<'PY'
import json, sys
g = sys.argv[1]
j = json.loads(open(f"gpurun_out/grid_{g}.json").read().strip().splitlines()[-1])
print(g, round(j["value"], 2), round(j["roofline"]["frac"], 4), j["roofline"]["per_iteration_ms"])
PY
done
