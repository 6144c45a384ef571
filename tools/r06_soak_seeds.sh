#!/bin/bash
# round 6: seed 21054 of the soak (quadratic, n = 2, m = 7, OWL-QN, vector-free extension) with and without the trial-side update
mkdir -p gpurun_out
for tu in 1 0; do
  echo "== LBFGS_OWL_TRIAL_UPDATE=$tu"
  LBFGS_OWL_TRIAL_UPDATE=$tu LBFGS_VF_TRACE=1 python tools/fuzz_soak.py --seeds 21054 2>&1 | grep -v "^coverage" | cut -c1-900
done
echo "== the two logistic seeds with the refined quotients (default build now)"
python tools/fuzz_soak.py --seeds 19077,24054 2>&1 | grep -v "^coverage" | cut -c1-300
