#!/bin/bash
O=gpurun_out
echo "== librto.so as built (no scratch in the denoise kernels)" > $O/r3_contention_determinism.txt
bash tools/contention_check.sh 150 >> $O/r3_contention_determinism.txt
echo "== the same sources with -DRTO_NET_SQ0_WG=4 (the all-planes GuidanceNet instantiations spill 12-20 B per lane to scratch)" >> $O/r3_contention_determinism.txt
RTO_LIB=$PWD/rt-octree_amd/lib_ab/librto_1.so bash tools/contention_check.sh 150 >> $O/r3_contention_determinism.txt
echo "== that build, ONE process on the GPU" >> $O/r3_contention_determinism.txt
RTO_LIB=$PWD/rt-octree_amd/lib_ab/librto_1.so python3 tools/contention_determinism.py 1 150 2>&1 | grep "^seed" >> $O/r3_contention_determinism.txt
cut -c1-330 $O/r3_contention_determinism.txt
