#!/bin/bash
# generic A/B of variant libraries with 200-step lines: tools/archive/r03_ab.sh <tag> "<bench args>;<bench args>;..." lib1 lib2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; TAG=$1; ARGSETS=$2; shift 2
mkdir -p gpurun_out/$TAG
IFS=';' read -ra SETS <<< "$ARGSETS"
for rep in 1 2; do
for args in "${SETS[@]}"; do
  echo "== $args (rep $rep)"
  for lib in "$@"; do
    RR_LIB_PATH=$(readlink -f $lib) timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 $args 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$lib', '%.2f M steps/s' % (d['value']/1e6), 'kernel_ms %.4f' % d['roofline']['kernel_ms'])
"
  done
done
done 2>&1 | tee gpurun_out/$TAG/ab.txt
