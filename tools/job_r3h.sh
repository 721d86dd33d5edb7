ROOT=$PWD
O=$ROOT/gpurun_out
timeout 900 python3 -m pytest tests/test_render_parity.py tests/test_guidance_fused.py tests/test_cli.py -x -q -m gpu > $O/r3h_pytest.txt 2>&1; tail -4 $O/r3h_pytest.txt
for A in "" "--compact-records"; do
timeout 600 python3 bench.py $A --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$A', 'value %.0f'%d['value'], 'traverse %.3f shade %.3f'%(r['avg_launch_ms'], r['shade_kernel_avg_launch_ms']), 'tree MB %.0f'%d['config']['tree_device_mb'], d['parity_spot']['mismatches'])
" >> $O/r3h_compact.txt
done
cat $O/r3h_compact.txt
