#!/bin/bash
# Where the time of a hand-off of the persistent two-loop kernel goes, and what variants of it do to whole iterations.
#   bash tools/handoff_trace.sh "r2 slim late early" "tr_r2 tr_slim tr_late tr_early"
# First list: untraced builds under tools/bin/variants/<name> ("main" = the in-tree library), compared by the kernel's
# HIP-event time inside bench.py.  Second list: builds with -DLH_RES_TRACE=1, which print the phase breakdown of the
# hand-off (workgroups 0 and G/2) when their context goes.  Sizes: config 5's optimiser side, config 2, config 3's m at
# 1e7 and the 8-GPU run's shard.
set -o pipefail
VARIANTS_PLAIN=$1
VARIANTS_TRACE=$2
mkdir -p gpurun_out
run() {  # name n m extra-args...
  local v=$1 n=$2 m=$3; shift 3
  if [ "$v" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v; fi
  timeout -k 10 200 python bench.py --dim $n --hist $m --no-cpu-baseline --no-vector-free --steps 100 --repeats 3 "$@" \
      > gpurun_out/ht.json 2> gpurun_out/ht.err || { tail -5 gpurun_out/ht.err; return 1; }
  python - "$n" "$m" "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ht.json").read())
r = j["roofline"]; tl = r.get("two_loop") or {}
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} {sys.argv[3]:>9}: {j['value']:8.1f} it/s  kernel {(r.get('avg_ms') or 0)*1e3:7.1f} us = {r.get('achieved') or 0:5.0f} GB/s ({(r.get('frac') or 0)*100:4.1f} %)  two-loop {tl.get('ms', 0):.3f} ms", flush=True)
PY
  grep -E "res-(trace|skew)" gpurun_out/ht.err || true
}
IFS=";" read -ra CFGS <<< "${CONFIGS:-3000000 6;10000000 7;12500000 10}"   # "n m" pairs
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  for rep in 1 2; do
    for v in $VARIANTS_PLAIN; do run $v $1 $2 || exit 1; done
  done
  for v in $VARIANTS_TRACE; do run $v $1 $2 --no-prof || exit 1; done
done
