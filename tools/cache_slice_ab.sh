#!/bin/bash
# Hybrid persistent two-loop kernel: how much of the HBM part of q should keep the default cache policy (the rest, like the
# history vectors, is streamed with `nt`)?  KNOB=LBFGS_HIP_RESIDENT_KEEP_MB: the same sweep for the history vectors kept around the turnaround (shards that fit the chip).   bash tools/cache_slice_ab.sh "100000000 50000000 25000000" "slice slice_alt" "0 16 32 64 128 200 100000"
mkdir -p gpurun_out
for n in $1; do
  for v in $2; do
    for mb in $3; do
      if [ "$v" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v; fi
      env ${KNOB:-LBFGS_HIP_RESIDENT_PLAIN_MB}=$mb timeout -k 10 200 python bench.py --dim $n --hist ${4:-10} --no-cpu-baseline --no-vector-free --steps 60 --repeats 2 > gpurun_out/cs.json 2> gpurun_out/cs.err || { tail -3 gpurun_out/cs.err; exit 1; }
      python - "$n" "$v" "$mb" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/cs.json").read()); r = j["roofline"]
print(f"n={sys.argv[1]:>9} {sys.argv[2]:>9} plain {sys.argv[3]:>6} MiB: {j['value']:8.1f} it/s  kernel {(r.get('avg_ms') or 0)*1e3:8.1f} us", flush=True)
PY
    done
  done
done
