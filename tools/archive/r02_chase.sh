#!/bin/bash
# contact-rich regime: bench lines (random + chase, both presets) of the product library, then the slow-wave collection (diagnostic build)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out/${1:-r02_chase}; mkdir -p $OUT
for P in T G; do
  timeout -k 10 200 python3 bench.py --preset $P --no-cpu-baseline > $OUT/bench_${P}_random.json 2> $OUT/bench_${P}_random.err || exit 1
  timeout -k 10 200 python3 bench.py --preset $P --policy chase --steps 200 --warmup 150 --no-cpu-baseline > $OUT/bench_${P}_chase.json 2> $OUT/bench_${P}_chase.err || exit 1
done
python3 - <<PY
import json
for p in "TG":
    for m in ("random", "chase"):
        d = json.loads(open("$OUT/bench_%s_%s.json" % (p, m)).read().strip().splitlines()[-1])
        print(p, m, "%.1f M env-steps/s" % (d["value"] / 1e6), "kernel_ms", round(d["roofline"]["kernel_ms"], 4))
PY
if [ -f roborugby_amd/variants/lib_prof.so ]; then
  export RR_NO_ORDER=1 RR_LIB_PATH=roborugby_amd/variants/lib_prof.so
  timeout -k 10 250 python3 tools/chase_monsters.py G 40 150 > $OUT/cm_G.txt && timeout -k 10 250 python3 tools/chase_monsters.py T 40 150 > $OUT/cm_T.txt
  cut -c1-400 $OUT/cm_G.txt $OUT/cm_T.txt
fi
