#!/bin/bash
# In-situ A/B at the 8-GPU run's per-rank shard size (n_local = 1.25e7, m = 10) on ONE GPU: library variants built by
# tools/build_variants.sh and/or environment knobs.   bash tools/shard_ab.sh "name=ENV1=v,ENV2=v;variantdir ..."
#   e.g.  bash tools/shard_ab.sh "base=;base uvnt=LBFGS_HIP_NT_THRESHOLD_MB=0;uvnt"
N=${SHARD_N:-12500000}
for spec in $1; do
  name=${spec%%=*}; rest=${spec#*=}; envs=${rest%%;*}; var=${rest##*;}
  envs=${envs//,/ }
  env $envs LBFGS_HIP_LIB_DIR=tools/bin/variants/$var timeout -k 10 200 python bench.py --dim $N --no-cpu-baseline --no-vector-free --steps 60 --repeats 3 > gpurun_out/sa.json 2> gpurun_out/sa.err || { tail -5 gpurun_out/sa.err; exit 1; }
  python - "$name" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/sa.json").read())
r = j["roofline"]
print(f"{sys.argv[1]:>14}: {j['value']:7.1f} it/s  step kernel {r['avg_ms']*1e3:6.1f} us = {r['achieved']:.0f} GB/s  two-loop {r["two_loop"]["ms"]:.3f} ms = {100*r["two_loop"]["frac"]:.1f} %  per-iter {r['per_iteration_ms']}")
PY
done
