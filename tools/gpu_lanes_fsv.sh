#!/bin/bash
# distinct Sv39 fuzz programs against the lanes-per-wave layout
cd $GRAFT_REPO_ROOT
O=gpurun_out/lanes_fsv; mkdir -p $O
for l in ${LANES:-4 8 16 64}; do
  timeout -k 10 250 python bench.py --workload fuzz_sv39 --harts ${HARTS:-25600} --lanes $l --steps 2 --warmup 1 --no-cpu-baseline --no-traced-leg > $O/l$l.json 2> $O/l$l.err || { echo "lanes $l failed"; tail -3 $O/l$l.err; exit 1; }
  python -c "import json; d=json.load(open('$O/l$l.json')); print($l, round(d['value'],3), round(d['ms_per_step'],1))"
done
