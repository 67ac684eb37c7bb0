#!/bin/bash
# The round's closing measurements (gpurun -- bash tools/final_sweeps.sh): one sweep.py line per point into gpurun_out/final/.
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; rm -rf $O; mkdir -p $O
run() { local name=$1; shift; echo "== $name" ; timeout -k 10 240 python tools/sweep.py "$@" > $O/$name.jsonl 2> $O/$name.err || { echo "FAILED $name"; tail -3 $O/$name.err; exit 1; }; cut -c1-330 $O/$name.jsonl; }
run compute --workload compute --harts 4096,16384,65536,262144 --iters 1 &&
run compute_sv39 --workload compute_sv39 --harts 4096,65536,262144 --iters 1 &&
run fuzz --workload fuzz --harts 65536 &&
run sv39_fuzz --workload sv39 --harts 102400 &&
run sv39_fuzz_predecoded_bodies --workload sv39x --harts 16384 &&
run rvv_random --workload rvv --harts 16384 --trace &&
run rvvk_vlen512 --workload rvvk --harts 16384 --iters 4 --trace --vlenb 64 &&
run rvvk_vlen256 --workload rvvk --harts 16384 --iters 4 --trace --vlenb 32 &&
timeout -k 10 240 python bench.py --workload sv39 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_sv39.json 2> $O/bench_sv39.err && cut -c1-700 $O/bench_sv39.json
