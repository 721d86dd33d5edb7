#!/bin/bash
# Round 6: does the persistent traversal kernel leave room for the other stream's shading / network / filter kernels?  Sweep of
# the traversal's workgroups per CU (tuning key blocks_per_cu; 0 = all 8) x streams through bench.py's headline pass.
# bash tools/ab_overlap.sh [rounds]   (GPU box, repo root)
R=${1:-1}
for r in $(seq 1 $R); do
  for cfg in "0 1" "0 2" "0 3" "7 2" "6 2" "6 3" "5 2" "5 3" "4 3"; do
    set -- $cfg
    T=""; [ "$1" != "0" ] && T="--tuning blocks_per_cu=$1"
    python3 bench.py --streams $2 $T --steps 6 --warmup 2 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --count-frames 0 --spot-pixels 16 --no-exact-pass --no-full-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); rf=d['roofline']
print('round $r  traversal workgroups/CU %s  streams %s: value %6.0f  single-stream %6.0f  traverse %.3f shade %.3f ms/100 frames  parity %s' % ('$1', '$2', d['value'], d.get('value_single_stream') or 0, rf['avg_launch_ms'], rf['shade_kernel_avg_launch_ms'], (d.get('parity_spot') or {}).get('mismatches')))"
  done
done
