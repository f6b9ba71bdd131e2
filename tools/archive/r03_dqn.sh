#!/bin/bash
# r03: fused DQN learn step -- tests, then config 5 (python -m roborugby_amd.dqn) fused vs PyTorch path, stage timings, kernel stats
TAG=${1:-r03_dqn}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_dqn_fused.py tests/test_dqn_agent.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1; rc=$?
tail -n 25 $OUT/pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python -m roborugby_amd.dqn --num-envs 65536 --steps 600 --log-every 0 --out $OUT/dqn_fused.json > $OUT/dqn_fused.log 2>&1 || { echo "dqn fused failed"; tail -5 $OUT/dqn_fused.log; exit 1; }
tail -1 $OUT/dqn_fused.log | cut -c1-400
timeout -k 10 300 python -m roborugby_amd.dqn --num-envs 65536 --steps 600 --log-every 0 --no-fused --out $OUT/dqn_torch.json > $OUT/dqn_torch.log 2>&1 || { echo "dqn torch failed"; tail -5 $OUT/dqn_torch.log; exit 1; }
tail -1 $OUT/dqn_torch.log | cut -c1-400
timeout -k 10 300 python tools/dqn_profile.py > $OUT/dqn_stage_timings.txt 2>&1 || { echo "dqn_profile failed"; tail -5 $OUT/dqn_stage_timings.txt; }
tail -12 $OUT/dqn_stage_timings.txt
exit $rc
