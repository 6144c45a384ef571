#!/bin/bash
# The RCCL leg's two launch forms on ONE GPU with a 1-rank communicator (LBFGS_FORCE_RCCL=1: ncclAllReduce is the identity there,
# RCCL launches no kernel for it): what the gated exchange's machinery -- a second stream, a gate and a post kernel per hand-off,
# the epochs, the slots -- costs next to the persistent kernel alone, and what the kernel-per-step form costs.
#   bash tools/rccl_one_rank.sh [n ...]  -> gpurun_out/rccl_one_rank.log
mkdir -p gpurun_out
out=gpurun_out/rccl_one_rank.log
: > $out
export LBFGS_FORCE_RCCL=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
port=29600
for n in ${@:-12500224 100000000}; do
  for form in "none" "rccl gated" "rccl per-step"; do
    port=$((port + 1))
    case "$form" in
      none) env -u LBFGS_FORCE_RCCL -u RANK -u WORLD_SIZE -u LOCAL_RANK timeout -k 10 200 python bench.py --dim $n --no-cpu-baseline --no-vector-free --repeats 6 > gpurun_out/rccl1_tmp.json 2> gpurun_out/rccl1_tmp.err;;
      "rccl gated") MASTER_PORT=$port LBFGS_HIP_RCCL_RESIDENT=1 timeout -k 10 200 python bench.py --_rank-mode --comm rccl --dim $n --no-cpu-baseline --no-vector-free --repeats 6 > gpurun_out/rccl1_tmp.json 2> gpurun_out/rccl1_tmp.err;;
      *) MASTER_PORT=$port LBFGS_HIP_RCCL_RESIDENT=0 timeout -k 10 200 python bench.py --_rank-mode --comm rccl --dim $n --no-cpu-baseline --no-vector-free --repeats 6 > gpurun_out/rccl1_tmp.json 2> gpurun_out/rccl1_tmp.err;;
    esac
    rc=$?
    if [ $rc -ne 0 ]; then echo "n=$n $form: failed rc=$rc" | tee -a $out; tail -5 gpurun_out/rccl1_tmp.err; [ $rc -ge 124 ] && exit 1; continue; fi
    cp gpurun_out/rccl1_tmp.json "gpurun_out/rccl_one_rank_n${n}_$(echo $form | tr ' ' '_').json"
    python - "$n" "$form" >> $out <<'PY'
import json, sys
j = json.loads(open("gpurun_out/rccl1_tmp.json").read().strip().splitlines()[-1])
r = j["roofline"]; p = r["per_iteration_ms"]; ci = j["config"].get("comm_info") or {}
print(f"n={int(sys.argv[1]):>9} {sys.argv[2]:14s} {j['value']:9.2f} it/s | kernel {r.get('kernel', '')[:30]:30s} {r['avg_ms'] * 1e3:8.1f} us | two-loop {p['two_loop']:.4f} ms, "
      f"all-reduce launches/it {p['allreduce_launches']:.1f} | exchanges per two-loop {ci.get('exchanges_per_two_loop')} of {ci.get('exchange_us_mean')} us")
PY
    tail -1 $out
  done
done
