#!/bin/bash
# Collects the rocprofv3 evidence for one workload: kernel trace + stats, then one PMC pass each
# for FETCH_SIZE and WRITE_SIZE, then condenses them into gpurun_out/round<round>_<workload>_*.
#   bash bench_util/profile_round.sh <round> <workload>
# <workload>: a bench.py workload, or extra:<case> for a case of bench_util/profile_extra.py.
set -e -o pipefail
ROUND=$1
WL=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=${WL#extra:}
O=$R/gpurun_out/prof_${NAME}
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
cd $R
if [ "$WL" != "$NAME" ]; then
  LONG="bench_util/profile_extra.py $NAME"
  SHORT="$LONG"
else
  LONG="bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline --headline-only"
  SHORT="bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --headline-only"
fi
# the same command without the profiler, in this lease: tells the box from the tracer when the
# trace's durations are compared with another run's HIP events
UNPROF=none
if [ "$WL" = "$NAME" ]; then
  timeout -k 10 300 python3 $LONG > $O/unprofiled.json 2> $O/unprofiled.err
  UNPROF=$O/unprofiled.json
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $LONG > $O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $SHORT > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $SHORT > $O/write.log 2>&1
python3 bench_util/summarize_profile.py $O/trace $O/fetch $O/write $R/gpurun_out/round${ROUND}_${NAME} $NAME $UNPROF > $O/summary.log 2>&1
tail -2 $O/trace.log
