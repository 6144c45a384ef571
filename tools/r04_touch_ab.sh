#!/bin/bash
# round 4: the L2 touch behind the hand-off's first poll (resident.h LH_RES_TOUCH) against the plain build, at the sizes of
# configs 5 / 2 and of the 8-GPU run's shard, then the traced build of the winner's candidate (skew, hand-off phases).
#   bash tools/r04_touch_ab.sh "main touch8 touch16" "tr_main tr_touch8"
mkdir -p gpurun_out
CONFIGS="${CONFIGS:-3000000 6;10000000 7;12500224 10}" bash tools/handoff_trace.sh "$1" "$2"
