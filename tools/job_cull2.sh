O=gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/r3_cull_full_pytest.txt 2>&1; grep -E "passed|failed" $O/r3_cull_full_pytest.txt | tail -1
timeout 600 python3 bench.py > $O/r3_cull_bench.json 2> $O/r3_cull_bench.err
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3_cull_bench.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value %.0f exact %.0f"%(d["value"], d["value_exact"]), d["reference_timer"], "trav %.3f shade %.3f prep %.3f marched %.3f"%(r["avg_launch_ms"], r["shade_kernel_avg_launch_ms"], r["thresholds_kernel_avg_launch_ms"], r["tiles_marched_frac"]), d["parity_spot"]["mismatches"], d["reference_loop"]["pipelined"]["wall_fps"], d["psnr"]["denoised_db"])
PY
for A in "--spp 1 --no-denoise" "--c4"; do timeout 600 python3 bench.py $A --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$A value %.0f'%d['value'], 'trav %.3f shade %.3f prep %.3f marched %.3f'%(r['avg_launch_ms'], r['shade_kernel_avg_launch_ms'], r['thresholds_kernel_avg_launch_ms'], r['tiles_marched_frac']), d['parity_spot']['mismatches'])"; done
