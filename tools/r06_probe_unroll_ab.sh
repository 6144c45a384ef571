#!/bin/bash
# round 6: the trial probe (2 reads, two hashes per element) in situ at shard sizes against its unroll (16-byte loads in flight per
# stream and thread: 2 / 4 / 8) and its address map (2 = super-chunks, 1 = plain grid stride) -- builds of tools/build_variants.sh
mkdir -p gpurun_out
for cfg in "12500224 10" "10000000 7" "100000000 10"; do
  set -- $cfg
  for rep in 1 2; do
    for v in base pu2 pu8 pm1; do
      export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v
      timeout -k 10 300 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --no-live-traffic --steps 80 --repeats 5 > gpurun_out/pu.json 2> gpurun_out/pu.err || { tail -5 gpurun_out/pu.err; exit 1; }
      python - "$1" "$2" "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/pu.json").read())
r = j["roofline"]; pi = r["per_iteration_ms"]; t = j["config"]["line_search_trials_per_step"]
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} {sys.argv[3]:>5}: {j['value']:8.2f} it/s  two-loop {pi['two_loop']*1e3:8.1f}  update {pi['history_update']*1e3:7.1f}  line_eval {pi['line_eval']*1e3:6.1f} us = {pi['line_eval']*1e3/t:6.1f} per probe ({t:.2f} trials)", flush=True)
PY
    done
  done
done
