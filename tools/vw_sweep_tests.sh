#!/bin/bash
# the parity / shortcut / adversarial GPU tests under every built lane width (RR_VW; a width a preset is not built for falls back to its default)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out/vw_sweep; mkdir -p $OUT
for VW in 4 8 16 32 64; do
  RR_VW=$VW timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_differential_adversarial.py tests/test_gpu_shortcuts.py tests/test_gpu_goal.py -m gpu -q -p no:cacheprovider -k "not rollout_in_one_launch" > $OUT/vw$VW.log 2>&1
  rc=$?; echo "RR_VW=$VW rc=$rc: $(tail -1 $OUT/vw$VW.log)"
  if [ $rc -ge 124 ]; then exit $rc; fi
done
