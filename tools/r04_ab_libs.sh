#!/bin/bash
# r04: A/B of tuning builds (tools/build_variant.sh) on the random-policy headline lines: tools/r04_ab_libs.sh <tag> "<bench args>" name1 name2 ...  (three interleaved rounds)
TAG=$1; ARGS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
for rnd in 1 2 3; do for P in G T; do for v in "$@"; do
  RR_LIB_PATH=$ROOT/roborugby_amd/variants/lib_$v.so timeout -k 10 240 python bench.py --preset $P --steps 100 --warmup 20 --no-cpu-baseline $ARGS > $OUT/${P}_${v}_$rnd.json 2> $OUT/err.txt || { echo "bench failed"; tail -5 $OUT/err.txt; exit 1; }
  python - $OUT/${P}_${v}_$rnd.json "round $rnd $P $v" <<'PY' | tee -a $OUT/ab.txt
import json, sys
d = json.load(open(sys.argv[1])); fr = d.get("from_reset") or {}
print("%s: %.1f M env-steps/s steady (kernel %.3f ms), %.1f M from reset, record %d B" % (sys.argv[2], d["value"] / 1e6, d["roofline"]["kernel_ms"], fr.get("value", 0) / 1e6, d["roofline"]["record_bytes_per_env"]))
PY
done; done; done
