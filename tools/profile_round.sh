#!/bin/bash
# The round's judged profile of `python bench.py` (n=1e8, m=10, one MI355X).  Run on the GPU box from the repo root:
#     bash tools/profile_round.sh r01
# The same at another size -- e.g. the per-rank shard of the 8-GPU run, where the two-loop is the resident kernel:
#     DIM=12500224 bash tools/profile_round.sh r04_shard_P8   (pmc_traffic.json -> profiles/pmc_traffic_shard_P8.json; the rank-0
#     shards of the metric are dist.shard_range(1e8, 0, P): 50000128, 25000192, 12500224 elements for P = 2, 4, 8)
# STEPS=<k> shortens the stats pass (default 400 iterations: ~5 s at n = 1e8).
# Writes gpurun_out/prof_<tag>/{bench.json, kernel_stats.csv, pmc_*_counter_collection.csv, pmc_traffic.json, summary.md}:
# copy them into profiles/ (pmc_traffic.json as profiles/pmc_traffic.json: bench.py's roofline.traffic reads it and names
# the raw CSVs it was derived from).  PMC counters are collected in their own passes, with --kernel-trace only.
# The stats pass is ONE long run (400 steps): the two-loop kernel of the first m iterations of a run has fewer steps
# (history not yet full), and with 10 such launches in 412 the plain average of kernel_stats.csv stays within ~1.5 % of the
# full-depth launches that bench.py times (summary.md lists the full-depth average separately).
set -e
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
# the build these passes are made with goes into pmc_traffic.json (tools/summarize_profile.py): bench.py's roofline.traffic_is_current
export LBFGS_HIP_BUILD_ID=$(python3 -c "import rust_lbfgs_amd as R; from rust_lbfgs_amd import _ffi; print(_ffi.load().lbfgs_hip_build_id().decode())")
dim=${DIM:-100000000}
hist=${HIST:-10}
if [ "$dim" = 100000000 ] && [ "$hist" = 10 ]; then size=""; else size="--dim $dim --hist $hist --no-cpu-baseline"; fi
python3 bench.py $size --no-live-traffic > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py $size --steps ${STEPS:-400} --repeats 1 --warmup 12 --no-cpu-baseline --no-vector-free > "$out/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $size --steps 4 --warmup 12 --no-cpu-baseline --no-prof --no-vector-free > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py $size --steps 4 --warmup 12 --no-cpu-baseline --no-prof --no-vector-free > "$out/pmc_write.log" 2>&1
LBFGS_PROFILE_COMMAND=bench.py python3 tools/summarize_profile.py "$out/stats" "$out/pmc_fetch" "$out/pmc_write" $dim "$out/summary.md" "$tag" $hist "$out/bench.json" > /dev/null
cp "$out"/stats/*/*_kernel_stats.csv "$out/kernel_stats.csv"
# the RAW counter rows behind roofline.traffic (one row per dispatch; a few hundred KB) are kept and committed
cat "$out"/pmc_fetch/*/*_counter_collection.csv > "$out/pmc_fetch_counter_collection.csv"
cat "$out"/pmc_write/*/*_counter_collection.csv > "$out/pmc_write_counter_collection.csv"
# keep only the small artefacts (the traces are hundreds of MB)
rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write"
cat "$out/summary.md"; cut -c1-200 "$out/bench.json"
