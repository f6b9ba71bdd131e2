#!/bin/bash
# occupancy probe: the SAME kernel source (preset T at 4 lanes per arena: 9.5 KB of LDS per wavefront, so LDS admits 4 waves per
# SIMD) compiled for 2, 3 and 4 waves per SIMD -- what does occupancy buy this kernel family, spills included?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_occ
mkdir -p $OUT
cd $ROOT
for n in 65536 262144; do for w in 2 3 4; do
  RR_LIB_PATH=$ROOT/roborugby_amd/variants/lib_occ$w.so timeout -k 10 200 python3 bench.py --preset T --arenas $n --steps 100 --warmup 20 --no-cpu-baseline 2>$OUT/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('T VW=%d arenas=$n waves/SIMD=$w' % d['config']['lanes_per_arena'], '%.1f M steps/s' % (d['value'] / 1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'])
" || { echo failed; tail -3 $OUT/err.txt; exit 1; }
done; done 2>&1 | tee $OUT/occ.txt
