#!/bin/bash
# round 6: launch grid of the accepted-step + history-update kernel (class 2, OpHistUpdateFromStep: 3r 4w + two hashes per element) and of
# the probe (class 4) in situ, now that the hashing is cheaper
mkdir -p gpurun_out
run() {  # run <dim> <hist> <label>
    timeout -k 10 300 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --no-live-traffic --steps 60 --repeats 5 > gpurun_out/ug.json 2> gpurun_out/ug.err || { tail -5 gpurun_out/ug.err; exit 1; }
    python - "$1" "$2" "$3" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ug.json").read())
r = j["roofline"]; pi = r["per_iteration_ms"]
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} {sys.argv[3]:>14}: {j['value']:8.2f} it/s  two-loop {pi['two_loop']*1e3:8.1f}  update {pi['history_update']*1e3:7.1f}  line_eval {pi['line_eval']*1e3:6.1f} us ({j['config']['line_search_trials_per_step']:.2f} trials)", flush=True)
PY
}
for cfg in "100000000 10" "12500224 10" "10000000 7"; do
  set -- $cfg
  for rep in 1 2; do
    for k2 in 0 27 40 48 96; do
      if [ "$k2" = 0 ]; then unset LBFGS_HIP_GRID_X32_K2; else export LBFGS_HIP_GRID_X32_K2=$k2; fi
      run $1 $2 "K2x32=$k2"
    done
    unset LBFGS_HIP_GRID_X32_K2
    if [ "$1" = 100000000 ]; then
      for k4 in 27 48 64 128; do
        LBFGS_HIP_GRID_X32_K4=$k4 run $1 $2 "K4x32=$k4"
      done
    fi
  done
done
