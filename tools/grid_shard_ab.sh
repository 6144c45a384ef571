#!/bin/bash
# launch grid A/B at a given n:  SHARD_N=3000000 bash tools/grid_shard_ab.sh "216 432 864"
N=${SHARD_N:-12500000}
for g in $1; do
  timeout -k 10 200 python bench.py --dim $N --no-cpu-baseline --no-vector-free --steps 60 --repeats 3 --grid $g > gpurun_out/ga.json 2> gpurun_out/ga.err || { tail -5 gpurun_out/ga.err; exit 1; }
  python - "$g" "$N" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ga.json").read())
r = j["roofline"]
print(f"n={sys.argv[2]} grid {sys.argv[1]:>5}: {j['value']:7.1f} it/s  step kernel {r['avg_ms']*1e3:6.1f} us = {r['achieved']:.0f} GB/s  two-loop {r['two_loop']['ms']:.3f} ms")
PY
done
