#!/bin/bash
# SQ instruction counters per launch of the small distinct-program batch at two layouts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_exp; rm -rf $O; mkdir -p $O
for L in 1 4; do
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/l$L -o p -- python3 $R/tools/exp_replicas.py 1600 16 $L > $O/out$L.jsonl 2> $O/err$L.log
cat $O/out$L.jsonl
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/l$L/**/p_counter_collection.csv", recursive=True)[0]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'wsp_step' not in r['Kernel_Name']: continue
    agg.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in agg.items():
    print(k, {c:int(x) for c,x in v.items()})
PY
done
find $O -name '*.db' -delete
