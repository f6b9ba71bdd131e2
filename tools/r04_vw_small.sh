#!/bin/bash
# r04: lanes per arena at BASELINE config 2's batch size (4,096 arenas leave most of the chip empty: 512 G wavefronts on 1,024 SIMDs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_vw_small; mkdir -p $OUT; cd $ROOT
for N in 4096 16384; do for cfg in "G 8" "G 16" "G 32" "G 64" "T 2" "T 4" "T 8" "T 64"; do
  set -- $cfg; P=$1; VW=$2
  RR_VW=$VW timeout -k 10 240 python bench.py --preset $P --arenas $N --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${P}_vw${VW}_$N.json 2> $OUT/err.txt || { echo "bench failed"; tail -5 $OUT/err.txt; exit 1; }
  python - $OUT/${P}_vw${VW}_$N.json "$N arenas, $P, $VW lanes per arena" <<'PY' | tee -a $OUT/lines.txt
import json, sys
d = json.load(open(sys.argv[1])); print("%s: %.1f M env-steps/s steady, %.1f M from reset (kernel %.3f ms)" % (sys.argv[2], d["value"] / 1e6, d["from_reset"]["value"] / 1e6, d["roofline"]["kernel_ms"]))
PY
done; done
