#!/bin/bash
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r3_k_pytest_gpu.txt 2>&1
tail -n 3 $OUT/r3_k_pytest_gpu.txt
bash tools/final_evidence.sh r3_k
