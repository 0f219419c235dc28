#!/bin/bash
# one bench line for workload $1 (headline only), phases printed
set -e
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --workload "$1" --headline-only > gpurun_out/r2_one.json 2>gpurun_out/r2_one.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2_one.json"))
print(d["ms_per_step"], d["phases_ms"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
