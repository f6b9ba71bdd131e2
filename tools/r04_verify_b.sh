#!/bin/bash
# r04 verification, part B: soaks of the exact shortcuts (on vs off, both builds, three shapes) and of the budgeted step (vs synchronous), round-4 build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_verify_b; mkdir -p $OUT; cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for ex in "" exact; do for P in G T D; do
  timeout -k 10 600 python tools/soak_shortcuts.py $P 16384 700 $ex 2>&1 | grep -v amdgpu.ids | tee -a $OUT/soak_shortcuts.txt
  [ ${PIPESTATUS[0]} -eq 0 ] || { echo "soak_shortcuts $P $ex failed"; exit 1; }
done; done
for spec in "T 16384 700 100000" "T 16384 700 20000" "G 8192 400 150000" "D 8192 400 80000"; do
  timeout -k 10 900 python tools/soak_budget.py $spec 2>&1 | grep -v amdgpu.ids | tee -a $OUT/soak_budget.txt
  [ ${PIPESTATUS[0]} -eq 0 ] || { echo "soak_budget $spec failed"; exit 1; }
done
