#!/bin/bash
# GPU parity tests, then (only if they are green) the bench line and the hart-count / lanes sweeps.  usage: gpu_check.sh <tag> [quick]
cd $GRAFT_REPO_ROOT
T=${1:-x}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -6 $O/tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit 1; fi
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err && cut -c1-330 $O/bench.json
timeout -k 10 300 python tools/sweep.py --harts 4096,16384,65536,262144 --lanes 0 --iters 1 > $O/sweep.jsonl 2> $O/sweep.err && cut -c1-250 $O/sweep.jsonl
timeout -k 10 300 python tools/sweep.py --harts 4096 --lanes 1,2,4,8,16,64 --iters 2 --no-check > $O/sweep_lpw.jsonl 2>> $O/sweep.err && cut -c1-250 $O/sweep_lpw.jsonl
timeout -k 10 200 python tools/sweep.py --workload fuzz --harts 65536 --lanes 0 > $O/fuzz.jsonl 2>> $O/sweep.err && cut -c1-200 $O/fuzz.jsonl
timeout -k 10 200 python tools/sweep.py --workload sv39 --harts 102400 --lanes 0 > $O/sv39.jsonl 2>> $O/sweep.err && cut -c1-200 $O/sv39.jsonl
timeout -k 10 200 python tools/sweep.py --workload rvv --harts 16384 --lanes 0 > $O/rvv.jsonl 2>> $O/sweep.err && cut -c1-200 $O/rvv.jsonl
