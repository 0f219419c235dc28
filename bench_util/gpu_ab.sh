#!/bin/bash
# A/B helper: bench lines (headline only, no CPU leg) for the workloads given, phases printed
#   bash bench_util/gpu_ab.sh <tag> <workload>...
set -e -o pipefail
TAG=$1
shift
mkdir -p gpurun_out
for wl in "$@"; do
  timeout -k 10 200 python bench.py --workload "$wl" --headline-only --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/ab_${TAG}_${wl}.json 2> gpurun_out/ab_${TAG}_${wl}.err
  python - "$TAG" "$wl" <<'PY'
import json, sys
tag, wl = sys.argv[1:3]
d = json.load(open("gpurun_out/ab_%s_%s.json" % (tag, wl)))
print(tag, wl, "step", d["ms_per_step"], "decoded", d.get("decoded_frac_of_hbm_peak"), d.get("phases_ms"), d.get("value"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"), flush=True)
PY
done
