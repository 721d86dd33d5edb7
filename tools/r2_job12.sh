#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_guidance_fused.py -m gpu -q 2>&1 | tail -2
for S in 1 2 3; do
python bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --streams $S --steps 512 --warmup 64 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams $S', 'fps %.0f'%d['value'], {k:round(v,4) for k,v in d['reference_timer'].items()}, 'trav %.3f'%d['roofline']['avg_launch_ms'])"
done
