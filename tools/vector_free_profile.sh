#!/bin/bash
# kernel times and HBM reads of the vector-free two-loop's kernels (rows, combine) at n = 1e8:  bash tools/vector_free_profile.sh
set -e
export TMPDIR=/tmp
out=gpurun_out/prof_vf; rm -rf $out gpurun_out/pmc_vf
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/vector_free_run.py > gpurun_out/prof_vf.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_vf -- python3 tools/vector_free_run.py > gpurun_out/pmc_vf.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof_vf/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_vf/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if len(v) > 3:
        print(k, len(v), "launches; HBM read (FETCH_SIZE x2)", round(max(v) * 1024 * 2 / 1e9, 3), "GB at most")
PY
rm -rf $out gpurun_out/pmc_vf
