#!/bin/bash
for r in 1 2 3; do
for f in rt-octree_amd/lib_ab/librto_0.so rt-octree_amd/lib_ab/librto_1.so; do
RTO_LIB=$PWD/$f timeout 300 python3 bench.py --c4 --steps 4 --warmup 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass --spot-pixels 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('c4 $f [$(cat ${f%/*}/flags_$(basename $f .so | sed s/librto_//).txt)] value %.0f persist %.3f shade %.3f thr %.3f'%(d['value'], r['avg_launch_ms'], r['shade_kernel_avg_launch_ms'], r['thresholds_kernel_avg_launch_ms']))"
done
done
