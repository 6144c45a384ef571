#!/bin/bash
# The host-closure bridge in isolation (tools/bridge_rate.hip): PCIe rates of the staging against plain hipMemcpy, and the
# drop-in lbfgs_minimize with a tight C closure, at n = 1e7 and 1e8.   bash tools/bridge_rate.sh  -> gpurun_out/bridge_rate.jsonl
set -e
mkdir -p gpurun_out tools/bin
lib=$(pwd)/rust-lbfgs_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/bridge_rate.hip -L "$lib" -llbfgs_solver -llbfgs_hip -Wl,-rpath,"$lib" -o tools/bin/bridge_rate
: > gpurun_out/bridge_rate.jsonl
for n in ${@:-10000000 100000000}; do
    timeout -k 10 300 tools/bin/bridge_rate $n 14 >> gpurun_out/bridge_rate.jsonl
done
cat gpurun_out/bridge_rate.jsonl
