#!/bin/bash
# the three tracked profiles of round 3 in one GPU call: headline bench, configs[4]'s shape, configs[3]'s shape
cd $GRAFT_REPO_ROOT
bash profiles/run_profile.sh r03_c > gpurun_out/prof_r03_c.log 2>&1 || { echo "r03_c failed"; tail -5 gpurun_out/prof_r03_c.log; exit 1; }
cut -c1-200 gpurun_out/prof_r03_c/bench.json
bash profiles/run_profile.sh r03_fuzz_sv39 --workload fuzz_sv39 --harts 25600 > gpurun_out/prof_r03_fuzz_sv39.log 2>&1 || { echo "fuzz_sv39 failed"; tail -5 gpurun_out/prof_r03_fuzz_sv39.log; exit 1; }
cut -c1-200 gpurun_out/prof_r03_fuzz_sv39/bench.json
bash profiles/run_profile.sh r03_rvv --workload rvv --harts 16384 > gpurun_out/prof_r03_rvv.log 2>&1 || { echo "rvv failed"; tail -5 gpurun_out/prof_r03_rvv.log; exit 1; }
cut -c1-200 gpurun_out/prof_r03_rvv/bench.json
