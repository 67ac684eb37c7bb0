#!/bin/bash
# parity tests, then the diagnostic lines of every workload.  usage: gpu_round3_check.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-x}; O=gpurun_out/$T; mkdir -p $O
bash tools/gpu_rvv_check.sh $T || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo BENCH FAILED; tail -3 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('compute', d['value'], d.get('value_traced'), d['config'].get('oracle_spot_check'))"
for h in 25600 102400; do
timeout -k 10 400 python bench.py --workload fuzz_sv39 --harts $h --steps 2 --warmup 1 --no-cpu-baseline --no-traced-leg > $O/fsv$h.json 2> $O/fsv$h.err || { echo FSV BENCH FAILED; tail -3 $O/fsv$h.err; exit 1; }
python -c "import json; d=json.load(open('$O/fsv$h.json')); print('fuzz_sv39', $h, d['value'], d['ms_per_step'], d['config'].get('oracle_spot_check'))"
done
timeout -k 10 200 python tools/sweep.py --workload fuzz --harts 65536 --lanes 0 > $O/fuzz.jsonl 2>> $O/sweep.err && cut -c1-200 $O/fuzz.jsonl
timeout -k 10 200 python tools/sweep.py --workload sv39 --harts 102400 --lanes 0 > $O/sv39.jsonl 2>> $O/sweep.err && cut -c1-200 $O/sv39.jsonl
timeout -k 10 200 python tools/sweep.py --workload rvv --harts 16384 --lanes 0 > $O/rvv.jsonl 2>> $O/sweep.err && cut -c1-200 $O/rvv.jsonl
timeout -k 10 200 python tools/sweep.py --workload compute_sv39 --harts 4096,65536 --lanes 0 > $O/csv39.jsonl 2>> $O/sweep.err && cut -c1-200 $O/csv39.jsonl
