#!/bin/bash
# kernel averages of one bench.py workload under rocprofv3's kernel trace:
#   bash bench_util/gpu_trace.sh <tag> <bench.py arguments...>
set -e -o pipefail
TAG=$1
shift
export TMPDIR=/tmp
rm -rf gpurun_out/trace_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_$TAG -- python3 bench.py "$@" --no-cpu-baseline > gpurun_out/trace_$TAG.log 2>&1
python3 - $(find gpurun_out/trace_$TAG -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-55s calls %5s avg %9.1f us total %8.2f ms' % (r['Name'].split('(')[0][-55:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
grep -o '"per_stripe_ms": {[^}]*}' gpurun_out/trace_$TAG.log || true
