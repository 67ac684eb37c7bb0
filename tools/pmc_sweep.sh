#!/bin/bash
# SQ counter passes over one tools/sweep.py point (general-path workloads): per simulated instruction figures for wsp_step_kernel.
# usage: bash tools/pmc_sweep.sh <tag> <sweep args...>      e.g.  pmc_sweep.sh rvv --workload rvv --harts 16384 --lanes 0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/pmcs_$T; rm -rf $O; mkdir -p $O
ARGS="--reps 1 --no-check $*"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -o p -- python3 $R/tools/sweep.py $ARGS > $O/a.json 2> $O/a.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -o p -- python3 $R/tools/sweep.py $ARGS > $O/b.json 2> $O/b.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC --output-format csv -d $O/c -o p -- python3 $R/tools/sweep.py $ARGS > $O/c.json 2> $O/c.err || echo "pass c failed"
python3 - <<PY
import csv, collections, json, os
line=json.loads(open("$O/a.json").read().strip().splitlines()[-1])
inst=line["instructions"]
out={}
for n in "abc":
    f=f"$O/{n}/p_counter_collection.csv"
    if not os.path.exists(f): continue
    agg=collections.defaultdict(float); disp=set()
    for r in csv.DictReader(open(f)):
        if "wsp_step_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    out.update({k: round(v/inst,3) for k,v in agg.items()})
    out["launches_"+n]=len(disp)
out["kernel_gips"]=line["kernel_gips"]; out["kernel_ms"]=line["kernel_ms"]; out["lanes_per_wave"]=line["lanes_per_wave"]
out["args"]="$*"
print(json.dumps(out))
open("$O/summary.json","w").write(json.dumps(out))
PY
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
