#!/bin/bash
# r04: config 5 with the learner on fewer than all CUs (RR_DQN_WGS) and a batch that divides evenly over them, so that the simulator's
# stragglers on the main stream find CUs of their own while the update kernels run on the side stream
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_dqn_wgs; mkdir -p $OUT; cd $ROOT
IFS=";" read -ra LIST <<< "${CFGS:-208 26624;240 30720;256 32768;256 32768 --no-overlap;256 32768;208 26624}"
for cfg in "${LIST[@]}"; do
  set -- $cfg; W=$1; B=$2; X=$3
  RR_DQN_WGS=$W timeout -k 10 300 python -m roborugby_amd.dqn --num-envs 65536 --steps ${STEPS:-1500} --log-every 0 --batch-size $B $X --out $OUT/dqn_${W}_${B}$X.json > $OUT/log.txt 2>&1 || { echo "failed"; tail -5 $OUT/log.txt; exit 1; }
  python - $OUT/dqn_${W}_${B}$X.json "wgs $W batch $B $X" <<'PY' | tee -a $OUT/lines.txt
import json, sys
d = json.load(open(sys.argv[1])); print("%s: %.2f M env-steps/s, %.2f samples per transition, mean step reward %.4f" % (sys.argv[2], d["env_steps_per_sec"] / 1e6, d["samples_per_transition"], d["mean_step_reward"]))
PY
done
