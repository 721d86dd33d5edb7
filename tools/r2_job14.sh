#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1800 python -m pytest tests -m gpu -q > $O/r2_j14_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j14_tests.log
grep -E "passed|failed|rc|FAILED|Error" $O/r2_j14_tests.log | tail -6
for B in 32 64; do
python bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --batch $B --steps 512 --warmup 64 2>&1 | grep -E '^\{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('batch $B', 'fps %.0f'%d['value'], {k:round(v,4) for k,v in d['reference_timer'].items()}, 'trav %.3f ms/launch'%d['roofline']['avg_launch_ms'], d['roofline']['frames_per_launch'])"
done
