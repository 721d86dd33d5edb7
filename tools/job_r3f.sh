ROOT=$PWD
O=$ROOT/gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/r3f_pytest.txt 2>&1
tail -4 $O/r3f_pytest.txt
timeout 600 python3 bench.py > $O/r3f_bench.json 2> $O/r3f_bench.err
tail -3 $O/r3f_bench.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r3f_bench_driver.json 2>> $O/r3f_bench.err
python3 - <<'PY'
import json
for f in ("gpurun_out/r3f_bench.json","gpurun_out/r3f_bench_driver.json"):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f,e); continue
    print(f, "value %.0f exact %.0f ms/step %.3f"%(d["value"], d.get("value_exact") or 0, d["ms_per_step"]))
    print(" ref_timer", d["reference_timer"])
    rl=d["reference_loop"]; print(" ref_loop fps %.0f wall %.0f pipelined %s"%(rl["fps"], rl["wall_fps"], rl.get("pipelined")))
    print(" parity", {k:v for k,v in (d["parity_spot"] or {}).items() if k!='what'})
    r=d["roofline"]; print(" roofline", {k:r[k] for k in ("achieved","frac","basis","traffic_stale","avg_launch_ms","shade_kernel_avg_launch_ms","thresholds_kernel_avg_launch_ms","algorithmic_frac")})
    print(" psnr", {k:v for k,v in (d["psnr"] or {}).items() if k not in ("note","hip_vs_cpu_oracle")})
    print(" cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"], d["config"]["tree_device_mb"])
PY
