#!/bin/bash
# the three tracked profiles of round 3 in one GPU call: headline bench, configs[4]'s shape, configs[3]'s shape.
# usage: gpu_profiles_r3.sh <headline tag> <fuzz tag> <rvv tag>
cd $GRAFT_REPO_ROOT
A=${1:-r03_c}; B=${2:-r03_fuzz_sv39}; C=${3:-r03_rvv}
bash profiles/run_profile.sh $A > gpurun_out/prof_$A.log 2>&1 || { echo "$A failed"; tail -5 gpurun_out/prof_$A.log; exit 1; }
cut -c1-200 gpurun_out/prof_$A/bench.json
bash profiles/run_profile.sh $B --workload fuzz_sv39 --harts 25600 > gpurun_out/prof_$B.log 2>&1 || { echo "$B failed"; tail -5 gpurun_out/prof_$B.log; exit 1; }
cut -c1-200 gpurun_out/prof_$B/bench.json
bash profiles/run_profile.sh $C --workload rvv --harts 16384 > gpurun_out/prof_$C.log 2>&1 || { echo "$C failed"; tail -5 gpurun_out/prof_$C.log; exit 1; }
cut -c1-200 gpurun_out/prof_$C/bench.json
