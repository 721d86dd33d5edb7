#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1800 python -m pytest tests -m gpu -q > $O/r2_j18_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j18_tests.log
grep -E "passed|failed|rc|FAILED|Error" $O/r2_j18_tests.log | tail -6
for E in "" "RTO_NO_SHREC=1"; do
env $E python bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 16 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$E', 'fps %.0f'%d['value'], {k:round(v,4) for k,v in d['reference_timer'].items()}, 'shade %.3f ms/launch'%d['roofline']['shade_kernel_avg_launch_ms'], 'ref_loop render %.4f'%d['reference_loop']['render_ms'], d['config']['tree_device_mb'])"
done
env python bench.py --c4 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --steps 3 --warmup 1 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c4', 'fps %.0f'%d['value'], {k:round(v,4) for k,v in d['reference_timer'].items()}, 'shade %.3f ms/launch'%d['roofline']['shade_kernel_avg_launch_ms'], d['config']['tree_device_mb'])"
