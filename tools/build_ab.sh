#!/bin/bash
# Same-box A/B of two builds of the whole tree: tools/bin/r04tree (round 4's HEAD; git-ignored) against the checked-out one.
# bench.py of each tree, alternating.   bash tools/build_ab.sh
# The old tree is made on the CPU box first:
#   git worktree add /tmp/r04tree 78b8ab8 && (cd /tmp/r04tree && python -c "import sys; sys.path.insert(0, '.'); import rust_lbfgs_amd as R; R.build()")
#   mkdir -p tools/bin/r04tree/profiles && cp -r /tmp/r04tree/{rust-lbfgs_amd,rust_lbfgs_amd.py,bench.py,include} tools/bin/r04tree/ && git worktree remove /tmp/r04tree --force
mkdir -p gpurun_out
out=gpurun_out/build_ab.log
: > $out
run() {  # run <label> <tree> <dim> <hist>
    ( cd $2 && timeout -k 10 150 python bench.py --dim $3 --hist $4 --repeats 8 --no-cpu-baseline --no-vector-free ) > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$1 failed"; tail -3 gpurun_out/ab_tmp.err; return 1; }
    python - "$1" $3 $4 >> $out <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
r = j["roofline"]; p = r["per_iteration_ms"]
print(f"n={int(sys.argv[2]):>9} m={sys.argv[3]:>2} {sys.argv[1]:8s} {j['value']:9.2f} it/s | kernel {r['avg_ms'] * 1e3:8.1f} us ({r['frac'] * 100:.1f} %) | two-loop {p['two_loop']:.4f} update {p['history_update']:.4f} line search {p['line_eval']:.4f} ms | build {r.get('loaded_build_id')}")
PY
    tail -1 $out
}
for cfg in "10000000 7" "12500224 10" "3000000 6" "100000000 10"; do
    set -- $cfg
    for rep in 1 2; do
        run r04 tools/bin/r04tree $1 $2 || exit 1
        run r05 . $1 $2 || exit 1
    done
done
