#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash tools/profile_round.sh r2_h c2 c5 c4 > $O/r2_h_profile.log 2>&1; tail -2 $O/r2_h_profile.log
timeout 1500 python bench.py --scenes 8 --scene-map both --steps 16 --warmup 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 > $O/r2_h_bench_c3_8scenes_1gpu.json 2> $O/r2_h_bench_c3.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2_h_bench_driver_cmd.json 2> $O/r2_h_bench_driver_cmd.err
for f in $O/r2_h_bench_c2.json $O/r2_h_bench_c5.json $O/r2_h_bench_c4.json $O/r2_h_bench_c3_8scenes_1gpu.json $O/r2_h_bench_driver_cmd.json; do grep '^{' $f | cut -c1-170; done
