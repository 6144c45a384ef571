#!/bin/bash
# EXPERIMENT ("Limit"): two independent single-rank optimisations on ONE GPU at the same time, each with the
# persistent two-loop kernel asking for every CU.  Short hand-off timeout so that a starved kernel ends quickly.
#   bash tools/two_processes_one_gpu.sh [n] [resident 0/1]
n=${1:-12500000}; r=${2:-1}
export LBFGS_HIP_HANDOFF_TIMEOUT_MS=${TIMEOUT_MS:-500} LBFGS_HIP_RESIDENT=$r
for k in 1 2; do
  ( timeout -k 10 120 python bench.py --dim $n --no-cpu-baseline --no-vector-free --steps 300 --repeats 1 > gpurun_out/two_$k.json 2> gpurun_out/two_$k.err; echo "process $k rc=$?" ) &
done
wait
for k in 1 2; do python - "$k" <<'PY'
import json, sys
k = sys.argv[1]
try:
    j = json.loads(open(f"gpurun_out/two_{k}.json").read())
    print(f"process {k}: {j['value']:.1f} it/s")
except Exception as e:
    print(f"process {k}: no result ({e}); stderr tail:", open(f"gpurun_out/two_{k}.err").read()[-300:].replace("\n", " | "))
PY
done
