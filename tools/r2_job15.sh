#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for B in 64 96 128; do
python bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --batch $B --steps 768 --warmup 128 2>&1 | grep -E '^\{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('batch $B', 'fps %.0f'%d['value'], {k:round(v,4) for k,v in d['reference_timer'].items()}, 'trav %.3f ms/launch'%d['roofline']['avg_launch_ms'], d['roofline']['frames_per_launch'])"
done
python -m pytest tests/test_render_parity.py -m gpu -q 2>&1 | tail -2
