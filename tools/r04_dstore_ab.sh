#!/bin/bash
# round 4: the direction's write-out pass of the persistent two-loop kernel with `nt` stores (as built) against the default
# cache policy (-DLH_RES_D_NT=0: the line search reads d next) -- whole iterations and per-class kernel times.
mkdir -p gpurun_out
for cfg in "3000000 6" "6000000 6" "10000000 7" "12500224 10"; do
  set -- $cfg
  for v in main dplain main dplain; do
    if [ "$v" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v; fi
    timeout -k 10 300 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --steps 100 --repeats 5 > gpurun_out/ds.json 2> gpurun_out/ds.err || { tail -5 gpurun_out/ds.err; exit 1; }
    python - "$1" "$2" "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ds.json").read())
r = j["roofline"]; pi = r["per_iteration_ms"]
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} {sys.argv[3]:>7}: {j['value']:8.1f} it/s  kernel {(r.get('avg_ms') or 0)*1e3:7.1f} us  two-loop {pi['two_loop']*1e3:7.1f}  update {pi['history_update']*1e3:6.1f}  line_eval {pi['line_eval']*1e3:6.1f} us ({j['config']['line_search_trials_per_step']:.2f} trials)", flush=True)
PY
  done
done
