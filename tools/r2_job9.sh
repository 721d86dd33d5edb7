#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests/test_bench_contract.py -m gpu -q 2>&1 | tail -3
timeout 1500 python bench.py --scenes 8 --scene-map both --steps 1600 --warmup 64 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 > $O/r2_c_bench_c3_8scenes_1gpu.json 2> $O/r2_c_bench_c3.err
tail -3 $O/r2_c_bench_c3.err; grep '^{' $O/r2_c_bench_c3_8scenes_1gpu.json | cut -c1-900
python bench.py > $O/r2_c_bench_c2.json 2> $O/r2_c_bench_c2.err; grep '^{' $O/r2_c_bench_c2.json | cut -c1-300
