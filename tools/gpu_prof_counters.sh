#!/bin/bash
# cycle accounting of the out-of-line paths on a -DWSP_PROF build (whisper_amd/lib/libwhisper_amd_prof.so)
cd $GRAFT_REPO_ROOT
export WSP_AMD_LIB=$PWD/whisper_amd/lib/libwhisper_amd_prof.so
O=gpurun_out/profc; mkdir -p $O
timeout -k 10 280 python tools/prof_run.py sv39v 25600 0 > $O/sv39v.json 2> $O/sv39v.err && cat $O/sv39v.json &&
timeout -k 10 200 python tools/prof_run.py rvvk 16384 0 > $O/rvvk.json 2> $O/rvvk.err && cat $O/rvvk.json &&
timeout -k 10 200 python tools/prof_run.py sv39 102400 0 > $O/sv39.json 2> $O/sv39.err && cat $O/sv39.json
