#!/bin/bash
# per-launch durations of the sliced run of a small distinct-program batch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt_exp; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/tools/exp_replicas.py 1600 16 ${1:-4} > $O/out.jsonl 2> $O/err.log
cat $O/out.jsonl
python3 - <<PY
import csv,glob
f=glob.glob("$O/kt/**/kt_kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'wsp_step' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows]
print(len(d), [round(x,1) for x in d])
PY
find $O -name '*.db' -delete
