#!/bin/bash
# r03 evidence for config 5: return curves over seeds (fused learn step vs the PyTorch path), the kernel micro-benchmark, rocprofv3 kernel stats
TAG=${1:-r03_dqn_ev}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for seed in 0 1 2; do
  timeout -k 10 400 python -m roborugby_amd.dqn --num-envs 65536 --steps 3000 --eval-every 300 --log-every 0 --seed $seed --out $OUT/dqn_T_65536_fused_seed$seed.json > $OUT/dqn_fused_seed$seed.log 2>&1 || { echo "fused seed $seed failed"; tail -5 $OUT/dqn_fused_seed$seed.log; exit 1; }
  python -c "
import json; d=json.load(open('$OUT/dqn_T_65536_fused_seed$seed.json')); print('fused seed $seed: %.1f M env-steps/s; greedy returns' % (d['env_steps_per_sec']/1e6), [round(c['greedy_return']) for c in d['curve'] if c['greedy_return'] is not None])"
done
for seed in 0 1; do
  timeout -k 10 600 python -m roborugby_amd.dqn --num-envs 65536 --steps 3000 --eval-every 300 --log-every 0 --seed $seed --no-fused --out $OUT/dqn_T_65536_torch_seed$seed.json > $OUT/dqn_torch_seed$seed.log 2>&1 || { echo "torch seed $seed failed"; tail -5 $OUT/dqn_torch_seed$seed.log; exit 1; }
  python -c "
import json; d=json.load(open('$OUT/dqn_T_65536_torch_seed$seed.json')); print('torch seed $seed: %.1f M env-steps/s; greedy returns' % (d['env_steps_per_sec']/1e6), [round(c['greedy_return']) for c in d['curve'] if c['greedy_return'] is not None])"
done
python tools/dqn_update_bench.py > $OUT/dqn_update_bench.txt 2>&1; tail -1 $OUT/dqn_update_bench.txt
python tools/dqn_profile.py > $OUT/dqn_stage_timings.txt 2>&1; tail -4 $OUT/dqn_stage_timings.txt
bash tools/dqn_kernel_stats.sh > $OUT/dqn_kernel_stats.txt 2>&1; head -12 $OUT/dqn_kernel_stats.txt
cp gpurun_out/dqn_stats/*/*_kernel_stats.csv $OUT/dqn_kernel_stats.csv 2>/dev/null
