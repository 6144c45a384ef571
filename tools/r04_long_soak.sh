#!/bin/bash
# round 4: long runs of the persistent two-loop on the final build -- 4e4 to 1e5 iterations per size, i.e. 0.5-2.6 million
# hand-offs each; any timed-out hand-off, fallback or backend error ends the run (bench.py exits non-zero).
mkdir -p gpurun_out
for cfg in "1200001 6 250" "10000000 7 100" "12500224 10 100"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --dim $1 --hist $2 --no-cpu-baseline --no-vector-free --steps 400 --repeats $3 > gpurun_out/soak.json 2> gpurun_out/soak.err || { echo "FAILED n=$1"; tail -5 gpurun_out/soak.err; exit 1; }
  python - "$1" "$2" "$3" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/soak.json").read())
c = j["config"]
print(f"n={sys.argv[1]:>9} m={sys.argv[2]:>2}: {int(sys.argv[3]) * 400} timed iterations in {sys.argv[3]} windows, {j['value']:.1f} it/s (median window); "
      f"restarts after convergence {c.get('restarts')}, two-loop = resident kernel: {(j['roofline'].get('two_loop') or {}).get('resident_kernel', j['roofline'].get('kernel'))}, "
      f"resident launches re-run per step: {(c.get('comm_info') or {}).get('resident_fallbacks')}", flush=True)
PY
done
