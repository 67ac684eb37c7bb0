set -e
cd $GRAFT_REPO_ROOT
for w in 2 4 8; do
  cp whisper_amd/lib/var_w$w.so whisper_amd/lib/libwhisper_amd.so
  for l in 1 2 4 8; do
    timeout -k 10 120 python bench.py --iters 4 --steps 2 --lanes $l --no-cpu-baseline > gpurun_out/sw_w${w}_l$l.json 2> gpurun_out/sw_w${w}_l$l.err
    python - <<PY
import json
d=json.load(open("gpurun_out/sw_w${w}_l$l.json"))
print("w$w lanes $l", round(d["roofline"]["kernel_gips"],3), d["config"]["oracle_spot_check"], d["config"]["all_harts_pass"])
PY
  done
done
