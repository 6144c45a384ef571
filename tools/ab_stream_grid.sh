# A/B: workgroups of the streaming kernels (update, probes) at the 8-GPU shard size; LBFGS_HIP_GRID overrides the default 216
for rep in 1 2; do
for g in default 128 256 432 512; do
  if [ $g = default ]; then unset LBFGS_HIP_GRID; else export LBFGS_HIP_GRID=$g; fi
  timeout -k 10 120 python bench.py --dim ${DIM:-12500224} --no-cpu-baseline --no-vector-free --no-live-traffic --repeats 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=r['per_iteration_ms']
print('grid=$g', round(d['value'],1), 'it/s | two_loop', round(p['two_loop']*1e3,1), 'update', round(p['history_update']*1e3,1), 'line_eval', round(p['line_eval']*1e3,1), 'trials', d['config']['line_search_trials_per_step'])"
done; done
