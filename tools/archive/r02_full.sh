#!/bin/bash
# r02 full evidence run from ONE build: GPU test-suite, bench lines, rocprofv3 kernel stats + HBM counters (G and T), SQ counters (G)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_full}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
bash tools/gpu_call.sh $TAG || { rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi; echo "(tests/bench returned $rc: continuing with the profiles)"; }
bash tools/profile_r01.sh ${TAG}_G > $OUT/prof_G.txt 2>&1 || { echo "profile G failed"; tail -5 $OUT/prof_G.txt; exit 1; }
bash tools/profile_r01.sh ${TAG}_T --preset T > $OUT/prof_T.txt 2>&1 || { echo "profile T failed"; tail -5 $OUT/prof_T.txt; exit 1; }
bash tools/pmc_sq2.sh ${TAG}_G > $OUT/sq_G.txt 2>&1 || { echo "sq G failed"; exit 1; }
tail -25 $OUT/sq_G.txt
