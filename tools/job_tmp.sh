#!/bin/bash
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
B="--steps 8 --warmup 2 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass --spot-pixels 0"
for i in 1 2; do
  for st in 1 2 3; do
    timeout 300 python bench.py $B --streams $st 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('streams $st value %.0f ms/step %.3f persist %.3f shade %.3f'%(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['shade_kernel_avg_launch_ms']))" | tee -a $OUT/streams_ab.txt
  done
done
