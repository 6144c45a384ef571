#!/bin/bash
# round 4, GPU batch C: calibration of the vector-free guard (cancellation figure on the six seeds of round 3's soaks and on
# healthy runs), the two-level hand-off in the stand-alone benchmark, vector-free tests.
mkdir -p gpurun_out
LBFGS_VF_TRACE=1 python tools/fuzz_soak.py --seeds 41623,83722,93075,96657,113165,191833 > gpurun_out/r04_vf_seeds.log 2> gpurun_out/r04_vf_seeds.err
grep -c EXACT gpurun_out/r04_vf_seeds.err; grep -E "^FAIL|^seeds" gpurun_out/r04_vf_seeds.log | cut -c1-300
awk '/vector-free/ {print $0}' gpurun_out/r04_vf_seeds.err | awk '{for(i=1;i<=NF;i++) if($i=="cancellation") print $(i+1), $NF}' | sort -g | awk '{a[NR]=$0} END {print "cancellation on the seeds: min", a[1], "| median", a[int(NR/2)+1], "| max", a[NR], "| n", NR}'
for cfg in "10000000 10" "1000000 6" "100000000 10"; do
  set -- $cfg
  LBFGS_VF_TRACE=1 timeout -k 10 300 python bench.py --dim $1 --hist $2 --no-cpu-baseline --steps 50 --repeats 1 > gpurun_out/vfb.json 2> gpurun_out/vfb.err || tail -n 5 gpurun_out/vfb.err
  echo "n=$1 m=$2: $(grep -c 'vector-free k=' gpurun_out/vfb.err) vector-free iterations, $(grep -c EXACT gpurun_out/vfb.err) redone exactly; cancellation min/median/max: $(grep 'vector-free k=' gpurun_out/vfb.err | sed 's/.*cancellation \([^ ]*\) .*/\1/' | sort -g | awk '{a[NR]=$1} END {print a[1], a[int(NR/2)+1], a[NR]}')"
  python -c "import json; j=json.load(open('gpurun_out/vfb.json')); print(j['value'], j['config']['extension_vector_free_two_loop'])"
done
tools/bin/handoff_bench 3000 > gpurun_out/r04_handoff_bench.log 2>&1; grep -E "=>|^----" gpurun_out/r04_handoff_bench.log | head -12
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "vector_free or gram or random_configurations" > gpurun_out/r04_gpu_suite_c.log 2>&1; tail -n 6 gpurun_out/r04_gpu_suite_c.log
