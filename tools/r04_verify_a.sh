#!/bin/bash
# r04 verification, part A (one build): GPU tests under every built lane width, wavefront run-time distributions of the RANDOM policy at
# the steady state (diagnostic build: VERDICT r3 item 5 asked for it) and of the chase policy, SQ counters of the headline command
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_verify_a; mkdir -p $OUT; cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/vw_sweep_tests.sh | tee $OUT/vw_sweep.txt
grep -q "failed\|error" $OUT/vw_sweep.txt && exit 1
D=$ROOT/roborugby_amd/variants/lib_diag.so
for spec in "G random" "T random" "G chase" "T chase"; do
  set -- $spec
  if [ $2 = random ]; then export RR_STAGGER=1; else unset RR_STAGGER; fi
  RR_LIB_PATH=$D timeout -k 10 400 python tools/chase_monsters.py $1 40 150 $2 0 > $OUT/waves_$1_$2.txt 2>&1 || { echo "chase_monsters $spec failed"; tail -5 $OUT/waves_$1_$2.txt; exit 1; }
  head -2 $OUT/waves_$1_$2.txt | tail -1 | cut -c1-300
done
unset RR_STAGGER
bash tools/pmc_sq2.sh r04_G > $OUT/sq_G.txt 2>&1 || { echo "sq G failed"; tail -5 $OUT/sq_G.txt; exit 1; }
tail -26 $OUT/sq_G.txt
