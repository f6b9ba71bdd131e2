#!/bin/bash
# config 5: GPU tests of the trainer, then the long run with the return curve (overlap on and off)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out/r02_dqn2; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_dqn_agent.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1; rc=$?; tail -8 $OUT/pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python3 -m roborugby_amd.dqn --num-envs 65536 --steps 3000 --eval-every 300 --log-every 300 --out $OUT/dqn_T_65536.json > $OUT/dqn.log 2>&1 || { tail -20 $OUT/dqn.log; exit 1; }
tail -4 $OUT/dqn.log
timeout -k 10 600 python3 -m roborugby_amd.dqn --num-envs 65536 --steps 1200 --no-overlap --log-every 0 --out $OUT/dqn_T_65536_serial.json > $OUT/dqn_serial.log 2>&1 || { tail -20 $OUT/dqn_serial.log; exit 1; }
tail -1 $OUT/dqn_serial.log
