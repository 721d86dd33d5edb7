#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/profile_round.sh r2_d c2 c5 c4 > $O/r2_d_profile.log 2>&1; tail -3 $O/r2_d_profile.log
timeout 1500 python bench.py --scenes 8 --scene-map both --steps 16 --warmup 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 > $O/r2_d_bench_c3_8scenes_1gpu.json 2> $O/r2_d_bench_c3.err
for f in $O/r2_d_bench_c2.json $O/r2_d_bench_c5.json $O/r2_d_bench_c4.json $O/r2_d_bench_c3_8scenes_1gpu.json; do grep '^{' $f | cut -c1-180; done
