#!/bin/bash
# round 4, last GPU call: the un-profiled bench lines once the counter passes of THIS build are committed (so that
# roofline.traffic_is_current is true in the records), and the multi-process GPU tests with the round-4 assertions.
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_n1e8_m10.json 2> gpurun_out/final/bench_n1e8_m10.err
for p in 8 4 2; do
  case $p in 8) dim=12500224;; 4) dim=25000192;; 2) dim=50000128;; esac
  python bench.py --dim $dim --no-cpu-baseline > gpurun_out/final/shard_P${p}_bench.json 2> gpurun_out/final/shard_P${p}.err
done
python - <<'PY'
import json
for f in ("bench_n1e8_m10", "shard_P8_bench", "shard_P4_bench", "shard_P2_bench"):
    j = json.loads(open(f"gpurun_out/final/{f}.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print(f, round(j["value"], 1), "it/s  kernel", round(r["avg_ms"] * 1e3, 1), "us  frac", round(r["frac"], 3), " traffic", r.get("traffic"),
          "current", r.get("traffic_is_current"), r.get("traffic_build_id"))
PY
LBFGS_HIP_RESIDENT_GRID=120 timeout -k 10 500 python bench.py --gpus 2 --device 0 --exclusive-device 1 > gpurun_out/final/bench_two_ranks_sharing_one_gpu.json 2> gpurun_out/final/two_ranks.err
python -c "
import json; j=json.load(open('gpurun_out/final/bench_two_ranks_sharing_one_gpu.json')); print(j['value'], j['config']['legs'])"
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_resident_recovery.py -x -q > gpurun_out/final/gpu_tests.log 2>&1; tail -n 6 gpurun_out/final/gpu_tests.log
