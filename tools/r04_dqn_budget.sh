#!/bin/bash
# r04: config 5 under the budgeted step (arenas over the budget report NOT_READY and finish in a later call; only completed transitions are stored and counted)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_dqn_budget; mkdir -p $OUT; cd $ROOT
for B in 0 0 60000 100000 150000 250000; do
  timeout -k 10 300 python -m roborugby_amd.dqn --num-envs 65536 --steps ${STEPS:-1500} --log-every 0 --budget $B --out $OUT/dqn_budget_$B.json > $OUT/log.txt 2>&1 || { echo "failed"; tail -5 $OUT/log.txt; exit 1; }
  python - $OUT/dqn_budget_$B.json "budget $B" <<'PY' | tee -a $OUT/lines.txt
import json, sys
d = json.load(open(sys.argv[1])); print("%s: %.2f M env-steps/s (completed transitions %d of %d rows), %.2f samples per transition, mean step reward %.4f" % (sys.argv[2], d["env_steps_per_sec"] / 1e6, d["transitions"], d["num_envs"] * d["steps"], d["samples_per_transition"], d["mean_step_reward"]))
PY
done
