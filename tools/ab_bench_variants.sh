#!/bin/bash
# Same-box A/B of lib_ab builds through bench.py itself (per-stage ms of the reference timer): tools/ab_bench_variants.sh [rounds]
cd "$(dirname "$0")/.."
D=rt-octree_amd/lib_ab
for r in $(seq 1 ${1:-2}); do
  for f in $D/librto_*.so; do
    i=${f##*_}; i=${i%.so}
    RTO_LIB=$PWD/$f timeout 300 python3 bench.py --streams 1 --steps 4 --warmup 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass --no-full-pass --spot-pixels 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
t=d['reference_timer']
print('round $r variant $i [$(cat $D/flags_$i.txt)]: value %.0f render %.4f net %.4f filter %.4f ms/frame'%(d['value'],t['render_ms'],t['torch_ms'],t['filter_ms']))
"
  done
done
