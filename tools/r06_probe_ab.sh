#!/bin/bash
# round 6: the trial probe in situ at shard sizes -- launch grid of the evaluation class (LBFGS_HIP_GRID_X32_K4) x the direction
# written with / without `nt` by the persistent two-loop kernel (tools/bin/variants/dplain: -DLH_RES_D_NT=0)
mkdir -p gpurun_out
for cfg in "12500224 10" "10000000 7"; do
  set -- $cfg
  for rep in 1 2; do
  for v in main dplain; do
    for x32 in 0 27 48 64 128; do
    if [ "$v" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v; fi
    if [ "$x32" = 0 ]; then unset LBFGS_HIP_GRID_X32_K4; else export LBFGS_HIP_GRID_X32_K4=$x32; fi
    timeout -k 10 300 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --no-live-traffic --steps 100 --repeats 5 > gpurun_out/pab.json 2> gpurun_out/pab.err || { tail -5 gpurun_out/pab.err; exit 1; }
    python - "$1" "$2" "$v" "$x32" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/pab.json").read())
r = j["roofline"]; pi = r["per_iteration_ms"]
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} {sys.argv[3]:>7} K4x32={sys.argv[4]:>3}: {j['value']:8.1f} it/s  kernel {(r.get('avg_ms') or 0)*1e3:7.1f} us  two-loop {pi['two_loop']*1e3:7.1f}  update {pi['history_update']*1e3:6.1f}  line_eval {pi['line_eval']*1e3:6.1f} us ({j['config']['line_search_trials_per_step']:.2f} trials)", flush=True)
PY
    done
  done
  done
done
