#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1800 python -m pytest tests -m gpu -q > $O/r2_j10_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j10_tests.log
grep -E "passed|failed|rc|FAILED|Error" $O/r2_j10_tests.log | tail -6
for A in "" "--fp32-maps" "--exact-filter"; do
python bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 $A 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$A', 'fps %.0f'%d['value'], {k:round(v,4) for k,v in d['reference_timer'].items()}, d['config'].get('maps'))"
done
