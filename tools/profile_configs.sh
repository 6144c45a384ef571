#!/bin/bash
# rocprofv3 evidence for BASELINE.json's other single-GPU configs (2: quadratic n=1e7 m=7; 3: OWL-QN logistic n=1e7 m=6, the
# orthantwise.rs:140-161 path; 5: damped L-BFGS on Lennard-Jones, 1e6 atoms, lbfgs.rs:664-689), in the format of
# tools/profile_round.sh: kernel statistics, raw PMC rows (FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only)
# and one table per config.  Run on the GPU box from the repo root:
#     bash tools/profile_configs.sh r04 "2 3 5"
# Writes gpurun_out/prof_<tag>_config<k>/{run.jsonl, kernel_stats.csv, pmc_*_counter_collection.csv, pmc_traffic.json, summary.md}.
set -e
tag=${1:-r04}
root=$(pwd)
export TMPDIR=/tmp
export LBFGS_HIP_BUILD_ID=$(python3 -c "import rust_lbfgs_amd as R; from rust_lbfgs_amd import _ffi; print(_ffi.load().lbfgs_hip_build_id().decode())")
for k in ${2:-2 3 5}; do
  case $k in
    2) n=10000000; m=7;;
    3) n=10000000; m=6;;
    5) n=3000000; m=6;;
    *) echo "unknown config $k"; exit 1;;
  esac
  out=$root/gpurun_out/prof_${tag}_config$k
  rm -rf "$out"; mkdir -p "$out"
  python3 tools/run_configs.py --only config$k > "$out/run.jsonl" 2> "$out/run.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 tools/run_configs.py --only config$k > "$out/stats.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 tools/run_configs.py --only config$k > "$out/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 tools/run_configs.py --only config$k > "$out/pmc_write.log" 2>&1
  LBFGS_PROFILE_COMMAND="tools/run_configs.py --only config$k" python3 tools/summarize_profile.py "$out/stats" "$out/pmc_fetch" "$out/pmc_write" $n "$out/summary.md" "${tag}_config$k" $m > /dev/null
  cp "$out"/stats/*/*_kernel_stats.csv "$out/kernel_stats.csv"
  cat "$out"/pmc_fetch/*/*_counter_collection.csv > "$out/pmc_fetch_counter_collection.csv"
  cat "$out"/pmc_write/*/*_counter_collection.csv > "$out/pmc_write_counter_collection.csv"
  rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write"
  echo "== config $k (build $LBFGS_HIP_BUILD_ID)"; cat "$out/summary.md"; cut -c1-400 "$out/run.jsonl"
done
