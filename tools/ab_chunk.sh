#!/bin/bash
# Same-box sweep of the traversal's dequeue granularity (tuning key refill = 1000 * tiles per dequeue + 832) through bench.py:
#   bash tools/ab_chunk.sh [rounds] -- [bench args]      e.g.  bash tools/ab_chunk.sh 2 -- --c4
R=${1:-2}; shift; [ "$1" = "--" ] && shift
for r in $(seq 1 $R); do
  for T in 832 1832 2832 3832 4832; do
    python3 bench.py --tuning refill=$T --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --streams 1 --no-exact-pass --no-full-pass --count-frames 0 \
      --spot-pixels 16 --steps 6 --warmup 2 --groups-per-step 1 "$@" 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); rf = d['roofline']
    print('round $r refill=$T %8.0f frames/s  traverse %.3f  shade %.3f ms per launch  parity mismatches %s' % (d['value'], rf['avg_launch_ms'], rf['shade_kernel_avg_launch_ms'], (d.get('parity_spot') or {}).get('mismatches')))
"
  done
done
