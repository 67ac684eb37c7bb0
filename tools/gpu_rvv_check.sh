#!/bin/bash
# GPU parity tests, then the vector microbenchmarks and the RVV diagnostic bench lines.  usage: gpu_rvv_check.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-x}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -6 $O/tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit 1; fi
timeout -k 10 300 python tools/vbench.py > $O/vbench.jsonl 2> $O/vbench.err || { echo VBENCH FAILED; tail -3 $O/vbench.err; exit 1; }
python -c "
import json
for l in open('$O/vbench.jsonl'):
    d=json.loads(l); print(d['kind'], d['ok'], round(d['gips'],3), round(d['us_per_inst_per_hart'],2))
"
for v in 64 32; do
timeout -k 10 300 python bench.py --workload rvv --harts 16384 --vlenb $v --steps 3 --warmup 1 --no-cpu-baseline > $O/rvv$v.json 2> $O/rvv$v.err || { echo RVV BENCH FAILED; tail -3 $O/rvv$v.err; exit 1; }
python -c "import json; d=json.load(open('$O/rvv$v.json')); print('rvv vlenb', $v, d['value'], d.get('value_traced'), d['config'].get('oracle_spot_check'))"
done
