#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1800 python -m pytest tests -m gpu -q > $O/r2_j16_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j16_tests.log
grep -E "passed|failed|rc|FAILED|Error" $O/r2_j16_tests.log | tail -6
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2_j16_bench_driver.json 2> $O/r2_j16_bench_driver.err ) 2>&1 | grep real
grep '^{' $O/r2_j16_bench_driver.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.0f'%d['value'], 'ms/step %.3f'%d['ms_per_step'], d['reference_timer'], 'frac',round(r['frac'],3), 'trav %.3f ms per %s frames'%(r['avg_launch_ms'], r['frames_per_launch']), d['config']['workload'][:200])
print(d['reference_loop'])"
