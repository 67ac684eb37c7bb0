#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun from the repo root):
#   bash profiles/run_profile.sh <tag> [bench.py arguments of a diagnostic workload, e.g. --workload fuzz_sv39 --harts 25600]  -> gpurun_out/prof_<tag>/*; then here: python profiles/summarize.py gpurun_out/prof_<tag> <tag> <commit>
# 1. plain bench (JSON line), 2. rocprofv3 kernel trace + stats of the same command, 3./4. PMC passes
# (FETCH_SIZE and WRITE_SIZE need separate runs: 3 + 2 TCC counters), 5. SQ instruction mix.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03_b}; shift; WL="$@"; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-traced-leg $WL"
timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 1 $WL > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/kt.json 2> $O/kt.err
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py $ARGS > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py $ARGS > $O/pmc_write.json 2> $O/pmc_write.err
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o p -- python3 $R/bench.py $ARGS > $O/pmc_sq.json 2> $O/pmc_sq.err
find $O -name '*.db' -delete; ls -R $O | head -50; du -sh $O
