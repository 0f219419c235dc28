#!/bin/bash
# SQ counters of one workload's kernels: bash bench_util/gpu_pmc.sh <tag> <workload | extra:case> (env passes through)
set -e -o pipefail
TAG=$1; WL=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_${TAG}
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
cd $R
if [ "${WL#extra:}" != "$WL" ]; then
  SHORT="bench_util/profile_extra.py ${WL#extra:}"
else
  SHORT="bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --headline-only"
fi
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/a -- python3 $SHORT > $O/a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $O/b -- python3 $SHORT > $O/b.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
for part in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(o + "/" + part + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES", "SQ_BUSY_CYCLES"):
                cnt[k] += 1
    for k, v in acc.items():
        n = max(cnt[k], 1)
        print(part, k, {c: round(x / n) for c, x in v.items()}, "launches", n)
PY
