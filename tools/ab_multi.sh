#!/bin/bash
# A/B of several builds of the library over the standard workload points: usage ab_multi.sh <tag> lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/abm_$T; rm -rf $O; mkdir -p $O
for L in "$@"; do
  export WSP_AMD_LIB=$GRAFT_REPO_ROOT/whisper_amd/lib/$L
  echo "== $L" | tee -a $O/out.jsonl
  for ARGS in "--workload compute --harts 4096,65536,262144 --iters 1" "--workload compute_sv39 --harts 65536 --iters 1" "--workload sv39 --harts 102400" \
              "--workload fuzz --harts 65536" "--workload rvv --harts 16384 --trace" "--workload rvvk --harts 16384 --iters 4 --trace"; do
    timeout -k 10 200 python tools/sweep.py $ARGS >> $O/out.jsonl 2>> $O/err || { echo "FAILED $L $ARGS"; tail -3 $O/err; exit 1; }
  done
done
python3 - <<PY
import json
for ln in open("$O/out.jsonl"):
    if ln.startswith("=="): print(ln.strip()); continue
    d = json.loads(ln); print("  %-14s %7d harts  %8.3f GIPS" % (d["workload"], d["harts"], d["kernel_gips"]))
PY
