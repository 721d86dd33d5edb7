#!/bin/bash
# The plain bench lines of an evidence bundle, against the counters already committed in profiles/pmc_traffic.json
# (tools/final_evidence.sh takes the counters; lines run BEFORE they are installed say traffic_stale): bash tools/bench_lines.sh TAG
T=${1:-r3_k}; O=gpurun_out; mkdir -p $O
timeout 600 python3 bench.py > $O/${T}_bench_c2.json 2> $O/${T}_bench_c2.err
timeout 600 python3 bench.py --spp 1 --no-denoise --cpu-frames 0 --psnr-frames 16 > $O/${T}_bench_c5.json 2> $O/${T}_bench_c5.err
timeout 900 python3 bench.py --c4 --cpu-frames 0 --psnr-frames 16 > $O/${T}_bench_c4.json 2> $O/${T}_bench_c4.err
timeout 900 python3 bench.py --scenes 8 --scene-map both --steps 16 --warmup 1 --groups-per-step 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass > $O/${T}_bench_c3_8scenes_1gpu.json 2> $O/${T}_bench_c3.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_cmd.json 2> $O/${T}_bench_driver_cmd.err
for f in c2 c5 c4 c3_8scenes_1gpu driver_cmd; do grep '^{' $O/${T}_bench_$f.json | cut -c1-150; done
