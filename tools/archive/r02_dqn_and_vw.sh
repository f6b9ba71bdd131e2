#!/bin/bash
# (1) config 5 at 65,536 arenas with learning and evaluation curve; (2) lanes-per-arena A/B lines incl. VW=64 (one wavefront per arena)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_dqn
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m roborugby_amd.dqn --num-envs 65536 --steps ${DQN_STEPS:-3000} --eval-every 300 --log-every 300 --out $OUT/dqn_T_65536.json > $OUT/dqn.log 2>&1 || { echo "dqn failed"; tail -20 $OUT/dqn.log; exit 1; }
tail -15 $OUT/dqn.log
line() { # preset vw arenas extra
  RR_VW=$2 timeout -k 10 200 python3 bench.py --preset $1 --arenas $3 --steps 100 --warmup 20 --no-cpu-baseline $4 2>$OUT/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(json.dumps({'preset': '$1', 'RR_VW': '$2', 'arenas': $3, 'extra': '$4', 'lanes_per_arena': d['config']['lanes_per_arena'], 'env_steps_per_sec': d['value'], 'kernel_ms': d['roofline']['kernel_ms']}))
" || { echo failed $1 $2 $3; tail -3 $OUT/err.txt; return 1; }
}
{
for vw in 2 4 8 64; do line T $vw 65536 "" || exit 1; done
for vw in 2 4; do line T $vw 262144 "" || exit 1; done
for vw in 8 16 32 64; do line G $vw 65536 "" || exit 1; done
line T 2 65536 "--fuse 10" ; line G 8 65536 "--fuse 10"
line G 8 4096 "" ; line T 2 4096 "--dtype f32"; line G 8 65536 "--dtype f32"
} | tee $OUT/vw_ab.jsonl
