#!/bin/bash
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_filter_cull.py tests/test_guidance_fused.py tests/test_filter_parity.py tests/test_cli.py -x -q -m gpu > $OUT/fc_pytest.txt 2>&1
tail -n 8 $OUT/fc_pytest.txt
B="--steps 6 --warmup 2 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --spot-pixels 0"
for i in 1 2; do
  for v in "off:--no-filter-cull" "both:"; do
    name=${v%%:*}; fl=${v#*:}
    timeout 300 python bench.py $B $fl 2>/dev/null | grep '^{' | tail -n 1 > $OUT/fe_${name}_$i.json
  done
done
python - <<'PY'
import json,glob
for k in ("off","both"):
    for f in sorted(glob.glob("gpurun_out/fe_%s_*.json"%k)):
        d=json.loads(open(f).read()); t=d["exact_route"]["reference_timer"]
        print(k, "%.0f exact %.0f"%(d["value"], d["value_exact"]), "exact route: render %.4f net %.4f filter %.4f"%(t["render_ms"],t["torch_ms"],t["filter_ms"]))
PY
