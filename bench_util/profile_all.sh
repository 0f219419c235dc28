#!/bin/bash
# every profile of a round: bash bench_util/profile_all.sh <round> [workloads...]
ROUND=$1
shift
LIST=${@:-direct_w17 delta_w10 delta_w16 mixed_short_runs boolean_mix strings_w6 extra:patched_base extra:offgrid extra:dictionary extra:byte_rle extra:scatter}
for wl in $LIST; do
  echo "== $wl"
  bash bench_util/profile_round.sh $ROUND $wl || { echo "profile of $wl failed"; tail -5 gpurun_out/prof_${wl#extra:}/*.log; exit 1; }
done
ls gpurun_out/round${ROUND}_*
