#!/bin/bash
# How far apart do the workgroups of the persistent two-loop kernel ARRIVE at a hand-off, as a function of the grid?
# (256 workgroups stride through a vector 1 MiB apart: every workgroup meets the same HBM channels in every round; other
# grids rotate.)  Needs the traced build tools/bin/variants/tr_main.   bash tools/resident_skew_probe.sh "256 248 240 216" 10000000 10
mkdir -p gpurun_out
for g in $1; do
  LBFGS_HIP_RESIDENT_GRID=$g LBFGS_HIP_LIB_DIR=tools/bin/variants/${4:-tr_main} timeout -k 10 200 python bench.py --dim $2 --hist $3 \
      --no-cpu-baseline --no-vector-free --steps 60 --repeats 2 --no-prof > gpurun_out/sk.json 2> gpurun_out/sk.err || { tail -3 gpurun_out/sk.err; exit 1; }
  echo "grid $g n=$2 m=$3: $(python -c "import json;print(round(json.load(open('gpurun_out/sk.json'))['value'],1))") it/s  $(grep res-trace gpurun_out/sk.err)"
done
