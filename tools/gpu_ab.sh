#!/bin/bash
# A/B of two builds of the library on the GPU box: parity tests with build B, then bench + hart-count sweep for both.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
A=$GRAFT_REPO_ROOT/whisper_amd/lib/libwhisper_amd.so
B=$GRAFT_REPO_ROOT/whisper_amd/lib/libwhisper_amd_skipuni.so
export WSP_AMD_LIB=$B
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests_B.log 2>&1 && echo TESTS_B_OK
tail -5 $O/tests_B.log
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_B.json 2> $O/bench_B.err && cat $O/bench_B.json | cut -c1-400
export WSP_AMD_LIB=$A
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_A.json 2> $O/bench_A.err && cat $O/bench_A.json | cut -c1-400
for L in A B; do
  if [ $L = A ]; then export WSP_AMD_LIB=$A; else export WSP_AMD_LIB=$B; fi
  timeout -k 10 300 python tools/sweep.py --harts 4096,16384,65536,262144 --lanes 0 --iters 1 > $O/sweep_$L.jsonl 2> $O/sweep_$L.err && cat $O/sweep_$L.jsonl
  timeout -k 10 300 python tools/sweep.py --harts 4096 --lanes 1,2,4,16 --iters 2 --no-check > $O/sweep_lpw_$L.jsonl 2>> $O/sweep_$L.err && cat $O/sweep_lpw_$L.jsonl
done
