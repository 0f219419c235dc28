#!/bin/bash
# the file workloads at full size, with the wall time each takes (writing the file included)
set -e -o pipefail
mkdir -p gpurun_out
for wl in "$@"; do
  t0=$(date +%s.%N)
  timeout -k 10 500 python bench.py --workload $wl --steps 10 --warmup 2 --cpu-seconds 4 > gpurun_out/t_$wl.json 2> gpurun_out/t_$wl.err || { tail -5 gpurun_out/t_$wl.err; exit 1; }
  t1=$(date +%s.%N)
  python - $wl $t0 $t1 <<'PY'
import json, sys
wl, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
d = json.load(open("gpurun_out/t_%s.json" % wl))
print(wl, "wall %.1f s" % (t1 - t0), d["value"], d["ms_per_step"], d.get("path"), d.get("decoded_frac_of_hbm_peak"),
      d.get("per_stripe_ms"), d.get("cpu_baseline"), d["config"]["stripes"], d.get("file_GBps"), flush=True)
PY
done
