#!/bin/bash
# A/B of several builds of the library on one workload point: usage ab_libs.sh <tag> "<sweep args>" lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
T=$1; ARGS=$2; shift; shift
O=gpurun_out/ab_$T; mkdir -p $O
for L in "$@"; do
  export WSP_AMD_LIB=$GRAFT_REPO_ROOT/whisper_amd/lib/$L
  echo "== $L" >> $O/out.jsonl
  timeout -k 10 200 python tools/sweep.py $ARGS >> $O/out.jsonl 2>> $O/err || { echo "FAILED $L"; exit 1; }
done
cut -c1-260 $O/out.jsonl
