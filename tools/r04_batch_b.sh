#!/bin/bash
# round 4, GPU batch B: the final touch form against the build without it; the six vector-free seeds of round 3's soaks with the
# guard; configs 2 / 3; the two-ranks-on-one-GPU rehearsal record; then the GPU suite.
mkdir -p gpurun_out
{
  for cfg in "3000000 6" "6000000 6" "10000000 7" "12500224 10" "100000000 10"; do
    set -- $cfg
    for v in notouch main notouch main; do
      if [ "$v" = main ]; then unset LBFGS_HIP_LIB_DIR; else export LBFGS_HIP_LIB_DIR=tools/bin/variants/$v; fi
      timeout -k 10 300 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --steps 100 --repeats 3 > gpurun_out/tb.json 2> gpurun_out/tb.err || { tail -5 gpurun_out/tb.err; exit 1; }
      python - "$1" "$2" "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/tb.json").read())
r = j["roofline"]; tl = r.get("two_loop") or {}
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2} {sys.argv[3]:>8}: {j['value']:8.1f} it/s  kernel {(r.get('avg_ms') or 0)*1e3:7.1f} us = {r.get('achieved') or 0:5.0f} GB/s ({(r.get('frac') or 0)*100:4.1f} %)  two-loop {tl.get('ms', 0):.3f} ms", flush=True)
PY
    done
  done
  unset LBFGS_HIP_LIB_DIR
} > gpurun_out/r04_touch_final.log 2>&1
cat gpurun_out/r04_touch_final.log
python tools/fuzz_soak.py --seeds 41623,83722,93075,96657,113165,191833 > gpurun_out/r04_vector_free_guard_seeds.log 2>&1; tail -n 12 gpurun_out/r04_vector_free_guard_seeds.log
python tools/run_configs.py --only config2 > gpurun_out/r04_configs_tmp.jsonl 2> gpurun_out/r04_configs_tmp.err
python tools/run_configs.py --only config3 >> gpurun_out/r04_configs_tmp.jsonl 2>> gpurun_out/r04_configs_tmp.err
cut -c1-700 gpurun_out/r04_configs_tmp.jsonl
LBFGS_HIP_RESIDENT_GRID=120 timeout -k 10 500 python bench.py --gpus 2 --device 0 --exclusive-device 1 > gpurun_out/r04_bench_two_ranks_sharing_one_gpu.json 2> gpurun_out/r04_bench_two_ranks.err; tail -n 12 gpurun_out/r04_bench_two_ranks.err; cut -c1-1500 gpurun_out/r04_bench_two_ranks_sharing_one_gpu.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_suite_b.log 2>&1; tail -n 25 gpurun_out/r04_gpu_suite_b.log
