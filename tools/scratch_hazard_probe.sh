#!/bin/bash
# 8 processes on one GPU per probe kind: bash tools/scratch_hazard_probe.sh [iters] -> gpurun_out/r3_scratch_hazard_probe.txt
O=gpurun_out; mkdir -p $O; touch $O/r3_scratch_hazard_probe.txt
for K in ${KINDS:--1 0 1 2 3 4 5 7}; do
  for i in 1 2 3 4 5 6 7 8; do
    timeout 900 python3 tools/scratch_hazard_probe.py $i $K ${1:-100} 2>&1 | grep "^probe" > $O/shp_$i.txt &
  done
  wait
  cat $O/shp_[1-8].txt | awk '{n+=$(NF-3); d+=$(NF-1); l=$0} END {sub(/ seed.*/,"",l); print l ": " n " of " d " iterations over 8 processes"}' >> $O/r3_scratch_hazard_probe.txt
  rm -f $O/shp_[1-8].txt
done
cat $O/r3_scratch_hazard_probe.txt
