#!/bin/bash
# tests of entry-less streams, then their timings with a per-kernel breakdown
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_no_entries.py tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2_ne.log 2>&1 || { tail -30 gpurun_out/r2_ne.log; exit 1; }
tail -2 gpurun_out/r2_ne.log
timeout -k 10 300 python bench_util/no_entries_probe.py
rm -rf /tmp/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench_util/no_entries_probe.py > /tmp/kt.log 2>&1
cp "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" gpurun_out/r2_noentries_stats.csv
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r2_noentries_stats.csv')):
    n = r['Name']
    if 'orcd::' in n:
        print(n.split('(')[0][:60].ljust(60), r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
