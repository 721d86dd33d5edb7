#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_guidance_fused.py tests/test_filter_parity.py tests/test_cli.py tests/test_c4_full_size.py -m gpu -q > $O/r2_j8_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j8_tests.log
grep -E "passed|failed|rc|FAILED|Error" $O/r2_j8_tests.log | tail -5
python bench.py --cpu-frames 0 --psnr-frames 0 > $O/r2_j8_bench.json 2> $O/r2_j8_bench.err; grep '^{' $O/r2_j8_bench.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('fps %.0f'%d['value'], d['reference_timer'], 'frac',r['frac'], r['basis'][:20], 'tcp', r['tcp'] and {k:v for k,v in r['tcp'].items() if k in ('frac','line_accesses_per_clk_per_cu','l1_hit_rate','l2_hit_rate')}, 'alg', r['algorithmic_frac'])"
