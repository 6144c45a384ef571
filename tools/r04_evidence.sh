#!/bin/bash
# round 4: the judged evidence, all taken with ONE build (its id goes into every pmc_traffic*.json):
#   bash tools/r04_evidence.sh main      -> gpurun_out/prof_r04                      (n = 1e8, m = 10: the metric's configuration)
#   bash tools/r04_evidence.sh shards "8 4"  -> gpurun_out/prof_r04_shard_P8, _P4   (rank-0 shards of the 8- and 4-GPU runs, alone)
#   bash tools/r04_evidence.sh shards "2"    -> gpurun_out/prof_r04_shard_P2
#   bash tools/r04_evidence.sh configs "2 3 5"
# then copy into profiles/ with tools/r04_collect.sh.
set -e
what=$1
case $what in
  main) bash tools/profile_round.sh r04 ;;
  shards)
    for p in $2; do
      case $p in 8) dim=12500224;; 4) dim=25000192;; 2) dim=50000128;; *) echo "P=$p?"; exit 1;; esac
      DIM=$dim bash tools/profile_round.sh r04_shard_P$p
    done ;;
  configs) bash tools/profile_configs.sh r04 "$2" ;;
  *) echo "usage: $0 main | shards \"8 4 2\" | configs \"2 3 5\""; exit 1 ;;
esac
