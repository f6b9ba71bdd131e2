#!/bin/bash
# r03 evidence from ONE build: rocprofv3 kernel stats + HBM counter passes of bench.py's default command (G and T), SQ counters (G),
# wavefront timelines of the chase policy synchronous vs budgeted (diagnostic build)
TAG=${1:-r03_ev}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
bash tools/archive/r03_profile.sh ${TAG}_G > $OUT/prof_G.txt 2>&1 || { echo "profile G failed"; tail -5 $OUT/prof_G.txt; exit 1; }
echo "profile G done"
bash tools/archive/r03_profile.sh ${TAG}_T --preset T > $OUT/prof_T.txt 2>&1 || { echo "profile T failed"; tail -5 $OUT/prof_T.txt; exit 1; }
echo "profile T done"
bash tools/pmc_sq2.sh ${TAG}_G --no-stagger > $OUT/sq_G.txt 2>&1 || { echo "sq G failed"; exit 1; }
tail -25 $OUT/sq_G.txt
D=$ROOT/roborugby_amd/variants/lib_diag.so
if [ -f $D ]; then
  for spec in "G 0 sync" "G 200000 budget200k" "T 0 sync" "T 100000 budget100k"; do
    set -- $spec
    RR_LIB_PATH=$D timeout -k 10 300 python tools/chase_monsters.py $1 40 150 chase $2 > $OUT/waves_$1_$3.txt 2>&1 || { echo "chase_monsters $spec failed"; tail -3 $OUT/waves_$1_$3.txt; }
    head -2 $OUT/waves_$1_$3.txt | tail -1
  done
fi
