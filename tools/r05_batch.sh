#!/bin/bash
# round-5 GPU batches: every step under its own timeout; a step that times out or is killed ends the batch (no further GPU
# step is started after one), an ordinary failure (a failing test) does not.   bash tools/r05_batch.sh <batch>
mkdir -p gpurun_out
step() {  # step <seconds> <log> <command...>
    local t=$1 log=$2; shift 2
    echo "== $* (limit ${t}s)" | tee -a gpurun_out/batch.log
    timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1
    local rc=$?
    echo "   rc=$rc" | tee -a gpurun_out/batch.log
    tail -n 4 "gpurun_out/$log" | cut -c1-400
    if [ $rc -ge 124 ]; then echo "   timed out / killed: stopping the batch" | tee -a gpurun_out/batch.log; exit 1; fi
}
case "$1" in
c)
    step 500 steplock_all.log python -m pytest tests/test_gpu_step_locked.py -m gpu -q -k "not metric_size and not shard_size"
    step 120 vf_alpha_diag.log python tools/vf_alpha_diag.py quadratic_m7
    step 600 suite_c.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_lj.py tests/test_c_caller.py tests/test_gpu_fullsize.py -m gpu -x -q
    step 200 config3.jsonl python tools/run_configs.py --only config3
    step 300 config5.jsonl python tools/run_configs.py --only config5
    ;;
d)
    step 300 steplock_m7.log python -m pytest tests/test_gpu_step_locked.py -m gpu -q -k "quadratic_m7 or rosenbrock"
    step 300 lj_tests.log python -m pytest tests/test_gpu_lj.py -m gpu -x -q
    for rep in 1 2; do
        for fused in 1 0; do
            LBFGS_HIP_LJ_FUSED_TRIAL=$fused step 300 config5_fused${fused}_$rep.jsonl python tools/run_configs.py --only config5
        done
    done
    step 300 small_vector_ab.log python tools/small_vector_ab.py
    ;;
*) echo "unknown batch"; exit 2;;
esac
