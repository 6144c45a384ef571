#!/bin/bash
# hipGraph replay of the two-loop vs eager launches, whole iterations:  bash tools/graph_ab.sh "100000 1000000 3000000 12500000" 10
for n in $1; do
  for g in 0 1 0 1; do
    LBFGS_HIP_GRAPH=$g timeout -k 10 200 python bench.py --dim $n --hist ${2:-10} --no-cpu-baseline --no-vector-free --no-prof --steps 60 --repeats 5 > gpurun_out/gr.json 2> gpurun_out/gr.err || { tail -5 gpurun_out/gr.err; exit 1; }
    python - "$n" "$g" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/gr.json").read())
print(f"n={sys.argv[1]:>9} graph={sys.argv[2]}: {j['value']:8.1f} it/s (best {j['config']['best_repeat_iters_per_sec']:8.1f})  {j['ms_per_step']*1e3:7.1f} us/iter  trials/iter {j['config']['line_search_trials_per_step']:.2f}")
PY
  done
done
