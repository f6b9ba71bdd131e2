#!/bin/bash
# r04: targeted GPU tests (pytest -k expression), then the synchronous chase lines and the stuck-arena phase split of the CURRENT library
# usage: tools/r04_quick.sh <tag> "<pytest -k expr>" [diag]
TAG=${1:-r04_quick}; KEXPR=${2:-}; DIAG=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
rc=0
if [ -n "$KEXPR" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=10 -k "$KEXPR" > $OUT/pytest.log 2>&1
  rc=$?
  tail -n 25 $OUT/pytest.log
  echo "pytest exit code $rc"
  if [ $rc -ge 124 ]; then echo "pytest hung or was killed: no further GPU step"; exit $rc; fi
fi
B="timeout -k 10 240 python bench.py"
run() { name=$1; shift; $B "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }; python - <<PY
import json; d=json.load(open("$OUT/$name.json")); print("$name: %.1f M env-steps/s, %.3f ms/step, kernel %s ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
}
run bench_G_f64_chase --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline || exit 1
run bench_T_f64_chase --preset T --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline || exit 1
run bench_G_f64 --steps 100 --warmup 20 --no-cpu-baseline || exit 1
run bench_T_f64 --preset T --steps 100 --warmup 20 --no-cpu-baseline || exit 1
if [ -n "$DIAG" ] && [ -f roborugby_amd/variants/lib_diag.so ]; then
  for i in 0 1 2; do RR_NO_MEMO=1 RR_LIB_PATH=roborugby_amd/variants/lib_diag.so timeout -k 10 120 python tools/stuck_arena_phases.py G $i >> $OUT/stuck_G.txt 2>&1 || exit 1; done
  RR_NO_MEMO=1 RR_LIB_PATH=roborugby_amd/variants/lib_diag.so timeout -k 10 120 python tools/stuck_arena_phases.py T 0 >> $OUT/stuck_T.txt 2>&1 || exit 1
  grep "step latency" $OUT/stuck_G.txt $OUT/stuck_T.txt
fi
exit $rc
