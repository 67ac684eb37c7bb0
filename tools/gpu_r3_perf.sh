#!/bin/bash
# one GPU call: microbenchmarks + a short bench with a given tag
T=${1:-x}
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 300 python tools/ubench.py > $O/ubench_$T.jsonl 2> $O/ubench_$T.err || { echo UBENCH FAILED; tail -5 $O/ubench_$T.err; exit 1; }
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$T.json 2> $O/bench_$T.err || { echo BENCH FAILED; tail -5 $O/bench_$T.err; exit 1; }
python - <<PY
import json
for l in open("$O/ubench_$T.jsonl"):
    d=json.loads(l); print(d["kind"], d["ok"], round(d["gips"],2), round(d["ns_per_step"],1))
d=json.load(open("$O/bench_$T.json")); print("bench", d["value"], d.get("value_traced"), d["config"]["oracle_spot_check"], d["roofline"]["kernel_gips"])
PY
