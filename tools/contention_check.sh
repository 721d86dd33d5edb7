#!/bin/bash
# 8 processes on one GPU, each checking that every stage of the path repeats bit for bit (tools/contention_determinism.py):
# bash tools/contention_check.sh [iters] -> gpurun_out/contention_determinism.txt
O=gpurun_out; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  timeout 900 python3 tools/contention_determinism.py $i ${1:-200} 2>&1 | grep "seed" > $O/cd_$i.txt &
done
wait
cat $O/cd_[1-8].txt | tee $O/contention_determinism.txt; rm -f $O/cd_[1-8].txt
