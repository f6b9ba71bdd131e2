#!/bin/bash
# r04: lanes per arena (RR_VW) under the chase and the random policy, preset T (default 2) and G (default 8)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_vw; mkdir -p $OUT; cd $ROOT
for cfg in "T 2" "T 4" "T 8" "G 8" "G 16"; do
  set -- $cfg; P=$1; VW=$2
  for POL in chase random; do
    if [ $POL = chase ]; then ARGS="--policy chase --steps 200 --warmup 150 --no-stagger"; else ARGS="--steps 100 --warmup 20"; fi
    RR_VW=$VW timeout -k 10 240 python bench.py --preset $P $ARGS --no-cpu-baseline > $OUT/${P}_vw${VW}_$POL.json 2> $OUT/err.txt || { echo "bench failed"; tail -5 $OUT/err.txt; exit 1; }
    python - $OUT/${P}_vw${VW}_$POL.json "$P, $VW lanes per arena, $POL" <<'PY' | tee -a $OUT/lines.txt
import json, sys
d = json.load(open(sys.argv[1])); print("%s: %.1f M env-steps/s (kernel %.3f ms)" % (sys.argv[2], d["value"] / 1e6, d["roofline"]["kernel_ms"]))
PY
  done
done
