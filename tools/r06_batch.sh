#!/bin/bash
# round-6 GPU batches: every step under its own timeout; a step that times out or is killed ends the batch (no further GPU
# step is started after one), an ordinary failure (a failing test) does not.   bash tools/r06_batch.sh <batch>
mkdir -p gpurun_out
step() {  # step <seconds> <log> <command...>
    local t=$1 log=$2; shift 2
    echo "== $* (limit ${t}s)" | tee -a gpurun_out/batch.log
    mkdir -p "$(dirname "gpurun_out/$log")"
    case "$log" in
        *.json|*.jsonl) timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2> "gpurun_out/$log.err";;  # (stdout is the record itself)
        *) timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1;;
    esac
    local rc=$?
    echo "   rc=$rc" | tee -a gpurun_out/batch.log
    tail -n 4 "gpurun_out/$log" | cut -c1-400
    if [ $rc -ge 124 ]; then echo "   timed out / killed: stopping the batch" | tee -a gpurun_out/batch.log; exit 1; fi
}
case "$1" in
e1)  # final build: the whole GPU suite, then the judged profile of `python bench.py` and of the 8-GPU shard
    step 900 r06_gpu_suite.log python -m pytest tests -m gpu -q
    step 420 profile_r06.log bash tools/profile_round.sh r06
    DIM=12500224 step 300 profile_r06_P8.log bash tools/profile_round.sh r06_shard_P8
    ;;
e2)
    DIM=25000192 step 300 profile_r06_P4.log bash tools/profile_round.sh r06_shard_P4
    DIM=50000128 step 300 profile_r06_P2.log bash tools/profile_round.sh r06_shard_P2
    step 600 profile_r06_configs.log bash tools/profile_configs.sh r06 "2 3 5"
    step 300 vf_profile.out bash tools/vector_free_profile.sh
    step 400 eight_ranks.err python tools/eight_ranks_one_gpu.py
    step 120 probe_alu_check.log python tools/probe_alu_check.py 12500224 10000000 100000000
    step 300 rccl_one_rank.out bash tools/rccl_one_rank.sh
    ;;
e3)  # after the counter passes of THIS build are in profiles/: the un-profiled lines (roofline.traffic_is_current = true)
    mkdir -p gpurun_out/final
    step 300 final/bench_n1e8_m10.json python bench.py
    for p in 8 4 2; do
        case $p in 8) dim=12500224;; 4) dim=25000192;; 2) dim=50000128;; esac
        step 200 final/shard_P${p}_bench.json python bench.py --dim $dim --no-cpu-baseline
    done
    # the launch form the RCCL leg takes without the gated exchange (a kernel per two-loop step), alone at the 8- and 4-GPU shard sizes: for the scaling model
    LBFGS_HIP_RESIDENT=0 step 200 final/shard_P8_per_step_bench.json python bench.py --dim 12500224 --no-cpu-baseline --no-vector-free
    LBFGS_HIP_RESIDENT=0 step 200 final/shard_P4_per_step_bench.json python bench.py --dim 25000192 --no-cpu-baseline --no-vector-free
    step 200 final/bench_3e6.json python bench.py --dim 3000000 --hist 6 --no-cpu-baseline
    step 200 config2.jsonl python tools/run_configs.py --only config2
    step 200 config3.jsonl python tools/run_configs.py --only config3
    step 300 config5.jsonl python tools/run_configs.py --only config5
    step 300 c_callers.log python -m pytest tests/test_c_caller.py -m gpu -q -s
    ;;
*) echo "unknown batch"; exit 2;;
esac
