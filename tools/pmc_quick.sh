#!/bin/bash
# Ad-hoc SQ counter passes over one bench configuration: per simulated instruction figures for wsp_step_kernel.
# usage: bash tools/pmc_quick.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/pmc_$T; rm -rf $O; mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -o p -- python3 $R/bench.py $ARGS > $O/a.json 2> $O/a.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -o p -- python3 $R/bench.py $ARGS > $O/b.json 2> $O/b.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_MISC --output-format csv -d $O/c -o p -- python3 $R/bench.py $ARGS > $O/c.json 2> $O/c.err || echo "pass c failed"
python3 - <<PY
import csv, collections, json, os
inst=json.load(open("$O/a.json"))["config"]["instructions_per_step_per_gpu"]
out={}
for n in "abc":
    f=f"$O/{n}/p_counter_collection.csv"
    if not os.path.exists(f): continue
    agg=collections.defaultdict(float); disp=set()
    for r in csv.DictReader(open(f)):
        if "wsp_step_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    out.update({k: round(v/len(disp)/inst,3) for k,v in agg.items()})
out["kernel_gips"]=json.load(open("$O/a.json"))["roofline"]["kernel_gips"]
out["args"]="$*"
print(json.dumps(out))
open("$O/summary.json","w").write(json.dumps(out))
PY
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
