set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=/tmp/pmc_run
rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY} --output-format csv -d $O/sq -- python3 bench_util/profile_extra.py inflate > $O/sq.log 2>&1
tail -3 $O/sq.log
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pmc_run/sq/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    if 'inflate' in k: agg[k + ' grid ' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
