#!/bin/bash
# Same-box A/B of librto.so builds on the BIT-EXACT filter route (filter_fused; bench.py's value_exact):
#   bash tools/ab_exact.sh [rounds] LIB [LIB ...]   -> per round and library: value, value_exact, the exact pass's filter ms per 100 frames
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    RTO_LIB=$PWD/$L python3 bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-full-pass --count-frames 0 --spot-pixels 16 \
      --streams 1 --steps 4 --warmup 1 --groups-per-step 2 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    er = d.get('exact_route') or {}
    print('round $r %-40s value %.0f value_exact %.0f  exact-route timer %s  parity mismatches %s' % ('$L', d['value'], d['value_exact'],
          {k: round(v * 100, 4) for k, v in (er.get('reference_timer') or {}).items() if isinstance(v, float)}, (d.get('parity_spot') or {}).get('mismatches')))
"
  done
done
