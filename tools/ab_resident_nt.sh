for rep in 1 2; do
for nt in default 4096; do
  if [ $nt = default ]; then unset LBFGS_HIP_RESIDENT_NT_MB; else export LBFGS_HIP_RESIDENT_NT_MB=$nt; fi
  timeout -k 10 120 python bench.py --dim 12500224 --no-cpu-baseline --no-vector-free --no-live-traffic --repeats 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=r['per_iteration_ms']
print('nt=$nt', round(d['value'],1), 'it/s | two_loop', round(p['two_loop']*1e3,1), 'update', round(p['history_update']*1e3,1), 'line_eval', round(p['line_eval']*1e3,1), 'trials', d['config']['line_search_trials_per_step'])"
done; done
