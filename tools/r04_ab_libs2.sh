#!/bin/bash
# r04: A/B of tuning builds on four lines (G / T x random steady state / synchronous chase), two interleaved rounds: tools/r04_ab_libs2.sh <tag> name1 name2 ...
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for rnd in 1 2; do for P in G T; do for POL in random chase; do
  if [ $POL = chase ]; then ARGS="--policy chase --steps 200 --warmup 150 --no-stagger"; else ARGS="--steps 100 --warmup 20"; fi
  for v in "$@"; do
    RR_LIB_PATH=$ROOT/roborugby_amd/variants/lib_$v.so timeout -k 10 240 python bench.py --preset $P $ARGS --no-cpu-baseline > $OUT/${P}_${POL}_${v}_$rnd.json 2> $OUT/err.txt || { echo "bench failed ($v)"; tail -5 $OUT/err.txt; exit 1; }
    python - $OUT/${P}_${POL}_${v}_$rnd.json "round $rnd $P $POL $v" <<'PY' | tee -a $OUT/ab.txt
import json, sys
d = json.load(open(sys.argv[1])); fr = d.get("from_reset") or {}
print("%s: %.1f M env-steps/s (kernel %.3f ms)%s" % (sys.argv[2], d["value"] / 1e6, d["roofline"]["kernel_ms"], (", %.1f M from reset" % (fr["value"] / 1e6)) if fr else ""))
PY
  done
done; done; done
