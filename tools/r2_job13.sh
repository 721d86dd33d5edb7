#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for B in 32 48 64; do
python bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --batch $B --steps 384 --warmup 64 2>&1 | grep -E '^\{|rror' | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t)
    print('batch $B', 'fps %.0f'%d['value'], {k:round(v,4) for k,v in d['reference_timer'].items()}, 'trav %.3f ms/launch'%d['roofline']['avg_launch_ms'], d['roofline']['frames_per_launch'])
except Exception as e: print('batch $B failed', t[:300])"
done
python -m pytest tests/test_render_parity.py -m gpu -q -x 2>&1 | tail -2
