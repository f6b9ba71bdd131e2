#!/bin/bash
# r03: budget sweep after pass-level parking
TAG=${1:-r03_budget2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_budget.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1; rc=$?
tail -n 12 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
B="timeout -k 10 240 python bench.py --no-cpu-baseline"
run() { name=$1; shift; $B "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }; python - <<PY
import json; d=json.load(open("$OUT/$name.json")); print("$name: %.1f M env-steps/s, %.3f ms/step, not_ready %.4f, kernel %s ms" % (d["value"]/1e6, d["ms_per_step"], d["config"].get("not_ready_fraction",0), d["roofline"]["kernel_ms"]))
PY
}
for b in 100000 150000 200000 250000 300000; do run G_budget_$b --policy chase --steps 200 --warmup 150 --no-stagger --budget $b || exit 1; done
for b in 30000 50000 100000 150000; do run T_budget_$b --preset T --policy chase --steps 200 --warmup 150 --no-stagger --budget $b || exit 1; done
run G_budget_200000_pipe2 --policy chase --steps 200 --warmup 150 --no-stagger --budget 200000 --pipeline 2 || exit 1
run T_budget_50000_pipe2 --preset T --policy chase --steps 200 --warmup 150 --no-stagger --budget 50000 --pipeline 2 || exit 1
