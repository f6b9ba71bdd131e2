#!/bin/bash
# r04 A/B: pass replay of the resolve loop (rr_sim.hpp: PassLog) on vs off (RR_NO_REPLAY=1), synchronous chase + random lines, three rounds interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04_replay_ab; mkdir -p $OUT; cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("%.1f M env-steps/s (kernel %.3f ms)" % (d["value"] / 1e6, d["roofline"]["kernel_ms"]))
PY
}
for rnd in 1 2 3; do
  for cfg in "G chase" "T chase" "G random" "T random"; do
    set -- $cfg; P=$1; POL=$2
    if [ $POL = chase ]; then ARGS="--preset $P --policy chase --steps 200 --warmup 150 --no-stagger --no-cpu-baseline"; else ARGS="--preset $P --steps 100 --warmup 20 --no-cpu-baseline"; fi
    for sw in on off; do
      if [ $sw = off ]; then export RR_NO_REPLAY=1; else unset RR_NO_REPLAY; fi
      timeout -k 10 240 python bench.py $ARGS > $OUT/${P}_${POL}_${sw}_$rnd.json 2> $OUT/err.txt || { echo "bench failed"; tail -5 $OUT/err.txt; exit 1; }
      echo "round $rnd $P $POL replay $sw: $(line $OUT/${P}_${POL}_${sw}_$rnd.json)" | tee -a $OUT/ab.txt
    done
  done
done
unset RR_NO_REPLAY
