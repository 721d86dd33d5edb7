#!/bin/bash
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^INFO: Scale" > $OUT/r3_l_pytest_gpu.txt
tail -n 12 $OUT/r3_l_pytest_gpu.txt
timeout 600 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --psnr-frames 8 2>/dev/null | grep '^{' | cut -c1-400
