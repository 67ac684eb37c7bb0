cd $GRAFT_REPO_ROOT; O=gpurun_out/exp2; mkdir -p $O
for L in main noscalar; do
  if [ $L = noscalar ]; then export WSP_AMD_LIB=$GRAFT_REPO_ROOT/whisper_amd/lib/libwhisper_amd_noscalar.so; fi
  timeout -k 10 200 python tools/sweep.py --harts 256,1024 --lanes 1 --iters 4 --no-check > $O/small_$L.jsonl 2> $O/err && cut -c1-200 $O/small_$L.jsonl
done
unset WSP_AMD_LIB
timeout -k 10 200 python tools/sweep.py --workload fuzz --harts 65536 --lanes 16,64 > $O/fuzz.jsonl 2>> $O/err && cut -c1-200 $O/fuzz.jsonl
timeout -k 10 200 python tools/sweep.py --workload fuzz --harts 4096 --lanes 1,4,16 > $O/fuzz4k.jsonl 2>> $O/err && cut -c1-200 $O/fuzz4k.jsonl
timeout -k 10 200 python tools/sweep.py --workload sv39 --harts 102400 --lanes 32,64 > $O/sv39.jsonl 2>> $O/err && cut -c1-200 $O/sv39.jsonl
timeout -k 10 200 python tools/sweep.py --workload rvv --harts 16384 --lanes 16,64 > $O/rvv.jsonl 2>> $O/err && cut -c1-200 $O/rvv.jsonl
