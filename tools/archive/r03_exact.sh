#!/bin/bash
# r03: the parity build (exact trig + scratch-rect carry) on the GPU -- parity tests with both libraries, then its cost on the bench line
TAG=${1:-r03_exact}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shortcuts.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1
rc=$?
tail -n 25 $OUT/pytest.log
if [ $rc -ge 124 ]; then echo "pytest hung or was killed: no further GPU step"; exit $rc; fi
for P in G T D; do
  for X in "" "--exact-trig"; do
    timeout -k 10 240 python bench.py --preset $P --steps 200 --warmup 20 --no-cpu-baseline $X > $OUT/bench_${P}${X:+_exact}.json 2> $OUT/bench_${P}${X:+_exact}.err || { echo "bench $P $X failed"; tail -5 $OUT/bench_${P}${X:+_exact}.err; exit 1; }
    python -c "import json; d=json.load(open('$OUT/bench_${P}${X:+_exact}.json')); print('$P $X: %.1f M env-steps/s, kernel %s ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
  done
done
exit $rc
