#!/bin/bash
# Kernel times and HBM bytes of the vector-free two-loop's kernels (rows, combine) on the CURRENT build:
#     bash tools/vector_free_profile.sh [n ...]          (default: 100000000 12500224; m = 10; one MI355X)
# Three rocprofv3 passes per size -- kernel statistics, FETCH_SIZE, WRITE_SIZE (PMC in passes of their own, --kernel-trace
# only) -- summarised against the algorithmic bytes of the 4m+3 passes into gpurun_out/vector_free_kernels.log
# (copy to profiles/rNN_vector_free_kernels.log).
set -e
export TMPDIR=/tmp
sizes=${@:-100000000 12500224}
log=gpurun_out/vector_free_kernels.log
mkdir -p gpurun_out
bid=$(python3 -c "import rust_lbfgs_amd as R; from rust_lbfgs_amd import _ffi; print(_ffi.load().lbfgs_hip_build_id().decode())")
echo "# bash tools/vector_free_profile.sh $sizes   (m = 10; build $bid; rocprofv3 kernel statistics + FETCH_SIZE + WRITE_SIZE passes)" > $log
for n in $sizes; do
    st=gpurun_out/prof_vf_$n; rm -rf $st ${st}_f ${st}_w
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $st -- python3 tools/vector_free_run.py $n > gpurun_out/prof_vf_$n.log 2>&1
    timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${st}_f -- python3 tools/vector_free_run.py $n > gpurun_out/pmc_vf_f_$n.log 2>&1
    timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${st}_w -- python3 tools/vector_free_run.py $n > gpurun_out/pmc_vf_w_$n.log 2>&1
    python3 - $n $st >> $log <<'PY'
import csv, glob, collections, sys
n, st, m = int(sys.argv[1]), sys.argv[2], 10
nb = 2 * m + 1
# algorithmic bytes: rows reads the 2m+1 basis vectors once (the three row vectors among them); combine reads the
# 2m+1 basis vectors and writes d
alg = {"gram_rows_resident_kernel": (8.0 * n * nb, 0.0), "gram_combine_resident_kernel": (8.0 * n * nb, 8.0 * n),
       "gram_rows_kernel": (8.0 * n * nb, 0.0), "OpGramCombine": (8.0 * n * nb, 8.0 * n)}
times = {}
for f in glob.glob(st + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        for k in alg:
            if k in r["Name"]:
                times[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
def pmc(dirn, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(dirn + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                for k in alg:
                    if k in r["Kernel_Name"]:
                        agg[k].append(float(r["Counter_Value"]))
    return agg
fetch, write = pmc(st + "_f", "FETCH_SIZE"), pmc(st + "_w", "WRITE_SIZE")
print(f"## n = {n}, m = {m}: 4m+3 = {4 * m + 3} passes = {8.0 * n * (4 * m + 3) / 1e9:.3f} GB per two-loop")
tot_us = 0.0
for k, (calls, us) in sorted(times.items()):
    rd, wr = alg[k]
    # gfx950: FETCH_SIZE counts 32-byte units in the 64-byte field -> x2 (MI355X_MICROARCH.md, HBM/rocprofv3 section); KB units
    f = sorted(fetch.get(k, []))
    w = sorted(write.get(k, []))
    fgb = f[len(f) // 2] * 1024 * 2 / 1e9 if f else float("nan")
    wgb = w[len(w) // 2] * 1024 / 1e9 if w else float("nan")
    rate = (rd + wr) / us / 1e6
    tot_us += us
    print(f"{k:32s} {calls:4d} launches  {us:9.1f} us  algorithmic {rd / 1e9:.3f} GB read + {wr / 1e9:.3f} GB written -> "
          f"{rate:.2f} TB/s = {rate / 8.0 * 100:.1f} % of 8 TB/s; PMC (median launch) {fgb:.3f} GB read + {wgb:.3f} GB written = "
          f"{(fgb + wgb) / ((rd + wr) / 1e9):.3f} x algorithmic")
print(f"rows + combine: {tot_us:.1f} us per two-loop (+ the scalar kernel)")
PY
    rm -rf $st ${st}_f ${st}_w
done
cat $log
