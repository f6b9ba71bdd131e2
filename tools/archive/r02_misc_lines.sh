#!/bin/bash
# the secondary bench lines DESIGN.md quotes: fp32 fast mode, config-[1] shapes (4,096 arenas), T at 262,144 arenas, fused rollout
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out/${1:-r02_misc}; mkdir -p $OUT
run() { tag=$1; shift; timeout -k 10 200 python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { echo "$tag failed"; tail -3 $OUT/bench_$tag.err; exit 1; }
  python3 -c "import sys,json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', '%.1f M env-steps/s' % (d['value']/1e6), 'ms_per_step %.4f' % d['ms_per_step'])"; }
run G_f32 --dtype f32 --steps 200 --warmup 20
run T_f32 --preset T --dtype f32 --steps 200 --warmup 20
run G_4096 --arenas 4096 --steps 300 --warmup 30
run T_4096_f32 --preset T --arenas 4096 --dtype f32 --steps 300 --warmup 30
run T_262144 --preset T --arenas 262144 --steps 100 --warmup 20
run T_fuse10 --preset T --fuse 10 --steps 200 --warmup 50
run G_fuse25 --fuse 25 --steps 100 --warmup 25
