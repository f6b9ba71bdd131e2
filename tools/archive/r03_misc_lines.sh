#!/bin/bash
# r03: the other BASELINE configurations and shapes as bench lines of the final build -> gpurun_out/<tag>/misc_lines/
TAG=${1:-r03_misc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG/misc_lines
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 240 python bench.py --no-cpu-baseline --steps 200 --warmup 20"
run() { name=$1; shift; $B "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }; python - <<PY
import json; d=json.load(open("$OUT/$name.json")); print("$name: %.1f M env-steps/s, %.4f ms/step, kernel %s ms, roofline frac %.4f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
}
run bench_G_4096_f32 --arenas 4096 --dtype f32 || exit 1          # BASELINE configs[1]: 4,096 arenas, fp32 state
run bench_T_4096_f32 --arenas 4096 --dtype f32 --preset T || exit 1
run bench_G_4096 --arenas 4096 || exit 1
run bench_T_4096 --arenas 4096 --preset T || exit 1
run bench_G_f32 --dtype f32 || exit 1
run bench_T_f32 --dtype f32 --preset T || exit 1
run bench_D_f64 --preset D || exit 1
run bench_T_262144 --preset T --arenas 262144 || exit 1
run bench_G_262144 --arenas 262144 || exit 1
run bench_T_fuse10 --preset T --fuse 10 --warmup 50 || exit 1
run bench_G_pipeline2 --pipeline 2 || exit 1
run bench_G_exact_trig --exact-trig || exit 1
run bench_T_exact_trig --exact-trig --preset T || exit 1
