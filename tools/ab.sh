#!/bin/bash
# A/B kernel variants: tools/ab.sh "<bench args>" lib1.so lib2.so ...   (prints env-steps/s and kernel ms per variant)
ARGS="$1"; shift
for lib in "$@"; do
  RR_LIB_PATH=$(readlink -f $lib) python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 $ARGS 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$lib', '%.3fM steps/s' % (d['value']/1e6), 'kernel_ms %.3f' % d['roofline']['kernel_ms'])
"
done
