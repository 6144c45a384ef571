#!/bin/bash
# vector-free (Gram) two-loop: library variants (tile sizes) inside bench.py at n=1e8, m=10:  bash tools/gram_ab.sh "g44 g28 old"
for v in $1; do
  LBFGS_HIP_LIB_DIR=tools/bin/variants/$v timeout -k 10 300 python bench.py --dim ${GRAM_N:-100000000} --hist ${GRAM_M:-10} --no-cpu-baseline --steps 30 --repeats 3 > gpurun_out/gm.json 2> gpurun_out/gm.err || { tail -5 gpurun_out/gm.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/gm.json").read())
e = j["config"]["extension_vector_free_two_loop"]["none"]
print(f"{sys.argv[1]:>6}: exact {j['value']:7.2f} it/s | vector-free {e['iters_per_sec']:7.2f} it/s  two-loop {e['two_loop_ms']:.3f} ms = {8*e['two_loop_passes']*j['config']['n']/e['two_loop_ms']/1e6:.0f} GB/s")
PY
done
