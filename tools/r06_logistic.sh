#!/bin/bash
# round 6: the hand-written logistic arithmetic against the ocml build (tools/bin/variants/ocml), grid sweep of the evaluation class
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r06_logistic_tests.log 2>&1 || { tail -30 gpurun_out/r06_logistic_tests.log; exit 1; }
tail -3 gpurun_out/r06_logistic_tests.log
{
echo "== new arithmetic, grid sweep (K4 = evaluation class)"
python tools/logistic_kernels.py 10000000 0 64 96 128 192
python tools/logistic_kernels.py 12500224 0 96 128 192
echo "== probe_alu_check (new)"
python tools/probe_alu_check.py 12500224 10000000 100000000
} > gpurun_out/r06_logistic_kernels_2.log 2>&1
cat gpurun_out/r06_logistic_kernels_2.log
{
echo "== config3 new"; python tools/run_configs.py --only config3
echo "== config3 new again"; python tools/run_configs.py --only config3
echo "== config2"; python tools/run_configs.py --only config2
} > gpurun_out/r06_config3_ab_2.log 2>&1
grep -o '"iters_per_sec": [0-9.]*' gpurun_out/r06_config3_ab_2.log
bash tools/profile_configs.sh r06 "3" > gpurun_out/r06_profile_config3.log 2>&1
cat gpurun_out/prof_r06_config3/summary.md | cut -c1-300
