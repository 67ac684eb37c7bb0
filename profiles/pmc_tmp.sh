set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_tmp; rm -rf $O; mkdir -p $O
ARGS="--iters 4 --steps 2 --warmup 1 --no-cpu-baseline --lanes ${LANES:-1}"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -o p -- python3 $R/bench.py $ARGS > $O/a.json 2> $O/a.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/b -o p -- python3 $R/bench.py $ARGS > $O/b.json 2> $O/b.err
python3 - <<PY
import csv, collections, json
inst=json.load(open("$O/a.json"))["config"]["instructions_per_step_per_gpu"]
for n in "ab":
    agg=collections.defaultdict(float); disp=set()
    for r in csv.DictReader(open(f"$O/{n}/p_counter_collection.csv")):
        if "wsp_step_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    print({k: round(v/len(disp)/inst,2) for k,v in agg.items()})
print(json.load(open("$O/a.json"))["roofline"]["kernel_gips"])
PY
find $O -name '*.csv' -delete
