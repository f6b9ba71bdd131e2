#!/bin/bash
# r03: wavefront run-time distribution of chase-policy launches, synchronous vs budgeted (diagnostic build), + bench lines
TAG=${1:-r03_timeline}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
D=$ROOT/roborugby_amd/variants/lib_diag.so
for spec in "G 0" "G 200000" "G 400000" "T 0" "T 100000"; do
  set -- $spec
  RR_NO_ORDER=${RR_NO_ORDER_DIAG:-0} RR_LIB_PATH=$D timeout -k 10 300 python tools/chase_monsters.py $1 40 150 chase $2 > $OUT/waves_$1_$2.txt 2>&1 || { echo "chase_monsters $spec failed"; tail -5 $OUT/waves_$1_$2.txt; exit 1; }
  head -2 $OUT/waves_$1_$2.txt
done
B="timeout -k 10 240 python bench.py --no-cpu-baseline"
run() { name=$1; shift; $B "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }; python - <<PY
import json; d=json.load(open("$OUT/$name.json")); print("$name: %.1f M env-steps/s, %.3f ms/step, not_ready %.4f, kernel %.3f ms" % (d["value"]/1e6, d["ms_per_step"], d["config"].get("not_ready_fraction",0), d["roofline"]["kernel_ms"]))
PY
}
run G_chase_sync --policy chase --steps 200 --warmup 150 --no-stagger || exit 1
for b in 200000 300000 400000; do run G_chase_budget_$b --policy chase --steps 200 --warmup 150 --no-stagger --budget $b || exit 1; done
run T_chase_sync --preset T --policy chase --steps 200 --warmup 150 --no-stagger || exit 1
for b in 50000 100000; do run T_chase_budget_$b --preset T --policy chase --steps 200 --warmup 150 --no-stagger --budget $b || exit 1; done
