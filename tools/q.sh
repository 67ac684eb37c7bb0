cd $GRAFT_REPO_ROOT; O=gpurun_out/q; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/t.log 2>&1; tail -4 $O/t.log
run() { local name=$1; shift; echo "== $name"; timeout -k 10 240 python tools/sweep.py "$@" > $O/$name.jsonl 2> $O/$name.err || { echo "FAILED $name"; tail -3 $O/$name.err; exit 1; }; cut -c1-60,140-290 $O/$name.jsonl; }
run compute --workload compute --harts 4096,65536 --iters 1 &&
run sv39_fuzz --workload sv39 --harts 102400 &&
run fuzz --workload fuzz --harts 65536 &&
run rvv --workload rvv --harts 16384 --trace
