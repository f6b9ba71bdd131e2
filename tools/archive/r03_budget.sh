#!/bin/bash
# r03: the budgeted step -- targeted GPU tests, then chase-policy bench lines at several budgets next to the synchronous ones.
# usage: tools/archive/r03_budget.sh <tag> ["pytest -k expr"]
TAG=${1:-r03_budget}; KEXPR=${2:-budget or thrust_entry or bench_gpus_2 or gloo_ranks}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=10 -k "$KEXPR" > $OUT/pytest.log 2>&1
rc=$?
tail -n 25 $OUT/pytest.log
echo "pytest exit code $rc"
if [ $rc -ge 124 ]; then echo "pytest hung or was killed: no further GPU step"; exit $rc; fi
B="timeout -k 10 240 python bench.py --no-cpu-baseline"
run() { name=$1; shift; $B "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }; python - <<PY
import json; d=json.load(open("$OUT/$name.json")); print("$name: %.1f M env-steps/s, %.3f ms/step, not_ready %.4f, kernel %.3f ms" % (d["value"]/1e6, d["ms_per_step"], d["config"].get("not_ready_fraction",0), d["roofline"]["kernel_ms"]))
PY
}
run G_random_stagger --steps 200 --warmup 20 || exit 1
run G_random_from_reset --steps 200 --warmup 20 --no-stagger || exit 1
run T_random_stagger --preset T --steps 200 --warmup 20 || exit 1
run T_random_from_reset --preset T --steps 200 --warmup 20 --no-stagger || exit 1
run G_chase_sync --policy chase --steps 200 --warmup 150 --no-stagger || exit 1
for b in 100000 200000 400000 800000; do run G_chase_budget_$b --policy chase --steps 200 --warmup 150 --no-stagger --budget $b || exit 1; done
run T_chase_sync --preset T --policy chase --steps 200 --warmup 150 --no-stagger || exit 1
for b in 50000 100000 200000 400000; do run T_chase_budget_$b --preset T --policy chase --steps 200 --warmup 150 --no-stagger --budget $b || exit 1; done
exit $rc
