#!/bin/bash
# Collects the rocprofv3 evidence for one workload: kernel trace + stats, then one PMC pass each
# for FETCH_SIZE and WRITE_SIZE, then condenses them into profiles/.
#   bash bench_util/profile_round.sh <round> <workload>
set -e -o pipefail
ROUND=$1
WL=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_${WL}
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
python3 bench_util/summarize_profile.py $O/trace $O/fetch $O/write $R/gpurun_out/round${ROUND}_${WL} $WL > $O/summary.log 2>&1
tail -3 $O/trace.log
