#!/bin/bash
# Build A/B variants of the HIP library with other compile-time traits for the dominant kernel, for in-situ comparison:
#   bash tools/build_variants.sh "1:1 1:2 1:3 1:4 2:2 2:4"      (MAP:UNROLL of OpTwoLoopStep)
# then e.g.  LBFGS_HIP_LIB_DIR=tools/bin/variants/m1u3 python bench.py ...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
# A second form names the variant and passes raw -D flags:  bash tools/build_variants.sh "b512=-DLH_BLOCK=512 nti=-DLH_NT_OUT=0u"
for v in $1; do
  if [[ $v == *=* ]]; then
    name=${v%%=*}; flags=${v#*=}; flags=${flags//,/ }   # commas separate several -D flags
    d=$root/tools/bin/variants/$name
  else
    m=${v%%:*}; u=${v##*:}
    d=$root/tools/bin/variants/m${m}u${u}
    flags="-DLH_STEP_MAP=$m -DLH_STEP_UNROLL=$u"
  fi
  # (same resource report + AGPR audit as the in-tree build; fails if the variant touches scratch memory)
  python "$root/rust-lbfgs_amd/_build.py" --variant "$(basename "$d")" $flags || exit 1
done
