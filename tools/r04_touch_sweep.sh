#!/bin/bash
# round 4: depth sweep of the touching wave (resident.h LH_RES_TOUCH_WAVE, LBFGS_HIP_RESIDENT_TOUCH = rounds) against the
# plain build and the all-waves form, per shard size.   bash tools/r04_touch_sweep.sh "0 4 8 12 16 24 32 48"
set -o pipefail
mkdir -p gpurun_out
run() {  # label libdir touch n m
  local label=$1 dir=$2 touch=$3 n=$4 m=$5
  if [ "$dir" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$dir; fi
  if [ -n "$touch" ]; then export LBFGS_HIP_RESIDENT_TOUCH=$touch; else unset LBFGS_HIP_RESIDENT_TOUCH; fi
  timeout -k 10 200 python bench.py --dim $n --hist $m --no-cpu-baseline --no-vector-free --steps 100 --repeats 3 \
      > gpurun_out/ts.json 2> gpurun_out/ts.err || { tail -5 gpurun_out/ts.err; return 1; }
  python - "$n" "$m" "$label" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ts.json").read())
r = j["roofline"]; tl = r.get("two_loop") or {}
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} {sys.argv[3]:>12}: {j['value']:8.1f} it/s  kernel {(r.get('avg_ms') or 0)*1e3:7.1f} us = {r.get('achieved') or 0:5.0f} GB/s ({(r.get('frac') or 0)*100:4.1f} %)  two-loop {tl.get('ms', 0):.3f} ms", flush=True)
PY
}
IFS=";" read -ra CFGS <<< "${CONFIGS:-3000000 6;6000000 6;10000000 7;12500224 10}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  run main main "" $1 $2 || exit 1
  for d in ${DEPTHS:-0 4 8 12 16 24 32 48}; do run "wave d=$d" tw $d $1 $2 || exit 1; done
  for v in ${ALLWAVES:-touch8 touch16}; do run $v $v "" $1 $2 || exit 1; done
  run main main "" $1 $2 || exit 1
done
