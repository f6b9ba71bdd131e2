#!/bin/bash
# r04: the fp32-state mode (RR_DTYPE_F32_STATE) -- its parity tests, then bench lines of the three precisions at BASELINE config 2's and the headline's size
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_fp32_state; mkdir -p $OUT; cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_fp32.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1; rc=$?
grep -E "^\[|passed|failed|Error|assert" $OUT/pytest.log | cut -c1-400
echo "pytest exit code $rc"
if [ $rc -ge 124 ]; then exit $rc; fi
for n in 4096 65536; do for P in G T; do for dt in f64 f32 f32_state; do
  timeout -k 10 240 python bench.py --preset $P --arenas $n --dtype $dt --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_${P}_${dt}_$n.json 2> $OUT/err.txt || { echo "bench failed"; tail -5 $OUT/err.txt; exit 1; }
  python - $OUT/bench_${P}_${dt}_$n.json "$P $dt $n" <<'PY' | tee -a $OUT/lines.txt
import json, sys
d = json.load(open(sys.argv[1])); print("%s: %.1f M env-steps/s steady, %.1f M from reset (kernel %.3f ms, record %d B)" % (sys.argv[2], d["value"] / 1e6, d["from_reset"]["value"] / 1e6, d["roofline"]["kernel_ms"], d["roofline"]["record_bytes_per_env"]))
PY
done; done; done
exit $rc
