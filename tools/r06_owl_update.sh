#!/bin/bash
# round 6: the OWL-QN trial that does the history update for its point -- parity, config 3 before/after is in the logs
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_step_locked.py -x -q > gpurun_out/r06_owl_update_tests.log 2>&1 || { tail -40 gpurun_out/r06_owl_update_tests.log; exit 1; }
tail -3 gpurun_out/r06_owl_update_tests.log
{
for i in 1 2 3; do python tools/run_configs.py --only config3; done
} > gpurun_out/r06_config3_trial_update.log 2>&1
grep -o '"iters_per_sec": [0-9.]*, "ms_per_iter": [0-9.]*, "trials_per_iter": [0-9.]*' gpurun_out/r06_config3_trial_update.log
bash tools/profile_configs.sh r06b "3" > gpurun_out/r06b_profile_config3.log 2>&1
cut -c1-330 gpurun_out/prof_r06b_config3/summary.md
