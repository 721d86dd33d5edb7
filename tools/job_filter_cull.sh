#!/bin/bash
# filter tile skip: parity tests, then an interleaved A/B of the headline bench with and without it
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
true

for i in 1 2 3; do
  timeout 300 python bench.py --steps 6 --warmup 2 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass --no-filter-cull 2>$OUT/fc_off_$i.err | tail -1 > $OUT/fc_off_$i.json
  timeout 300 python bench.py --steps 6 --warmup 2 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass 2>$OUT/fc_on_$i.err | tail -1 > $OUT/fc_on_$i.json
done
python - <<'PY'
import json,glob
for k in ("off","on"):
    for f in sorted(glob.glob("gpurun_out/fc_%s_*.json"%k)):
        try:
            d=json.loads(open(f).read())
            print(k, d["value"], d.get("ms_per_step"), d.get("reference_timer"), d.get("parity_spot"))
        except Exception as e:
            print(k, f, "unreadable", e)
PY
tail -5 $OUT/fc_on_1.err $OUT/fc_off_1.err
