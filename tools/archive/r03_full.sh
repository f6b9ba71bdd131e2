#!/bin/bash
# r03: whole GPU test-suite, then the bench lines of the round (synchronous + budgeted + pipelined), everything to gpurun_out/<tag>/
TAG=${1:-r03_full}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1100 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=15 > $OUT/pytest.log 2>&1
rc=$?
tail -n 30 $OUT/pytest.log
echo "pytest exit code $rc"
if [ $rc -ge 124 ]; then echo "pytest hung or was killed: no further GPU step"; exit $rc; fi
B="timeout -k 10 240 python bench.py"
run() { name=$1; shift; $B "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }; python - <<PY
import json; d=json.load(open("$OUT/$name.json")); print("$name: %.1f M env-steps/s, %.3f ms/step, not_ready %.4f, kernel %s ms" % (d["value"]/1e6, d["ms_per_step"], d["config"].get("not_ready_fraction",0), d["roofline"]["kernel_ms"]))
PY
}
run bench_G_f64 --steps 200 --warmup 20 || exit 1
run bench_G_f64_from_reset --steps 200 --warmup 20 --no-stagger --no-cpu-baseline || exit 1
run bench_T_f64 --preset T --steps 200 --warmup 20 --no-cpu-baseline || exit 1
run bench_T_f64_from_reset --preset T --steps 200 --warmup 20 --no-stagger --no-cpu-baseline || exit 1
run bench_G_f64_chase --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline || exit 1
run bench_G_f64_chase_budget --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline --budget 150000 || exit 1
run bench_G_f64_chase_budget_pipeline2 --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline --budget 150000 --pipeline 2 || exit 1
run bench_T_f64_chase --preset T --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline || exit 1
run bench_T_f64_chase_budget --preset T --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline --budget 100000 || exit 1
run bench_T_f64_chase_budget_pipeline2 --preset T --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline --budget 100000 --pipeline 2 || exit 1
exit $rc
