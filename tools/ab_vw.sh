#!/bin/bash
# A/B virtual-wave widths: tools/ab_vw.sh "<bench args>" vw1 vw2 ...
ARGS="$1"; shift
for vw in "$@"; do
  RR_VW=$vw python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 $ARGS 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('VW=$vw $ARGS', '%.3fM steps/s' % (d['value']/1e6), 'kernel_ms %.3f' % d['roofline']['kernel_ms'])
"
done
