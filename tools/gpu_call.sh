#!/bin/bash
# One gpurun call = one script: GPU test-suite, then (only if nothing hung) the bench lines.  Everything goes to gpurun_out/<tag>/.
# usage: tools/gpu_call.sh <tag> [pytest -k expression]
TAG=${1:-call}; KEXPR=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -n "$KEXPR" ]; then KARGS=(-k "$KEXPR"); else KARGS=(); fi
timeout -k 10 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=15 "${KARGS[@]}" > $OUT/pytest.log 2>&1
rc=$?
tail -n 40 $OUT/pytest.log
echo "pytest exit code $rc"
if [ $rc -ge 124 ]; then echo "pytest hung or was killed: no further GPU step"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > $OUT/bench_G_f64.json 2> $OUT/bench_G_f64.err || { echo "bench G failed"; tail -5 $OUT/bench_G_f64.err; exit 1; }
cat $OUT/bench_G_f64.json
timeout -k 10 300 python bench.py --preset T --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_T_f64.json 2> $OUT/bench_T_f64.err || { echo "bench T failed"; exit 1; }
cat $OUT/bench_T_f64.json
exit $rc
