#!/bin/bash
# Development A/B of the fused GuidanceNet kernel: in-tree build, then one rebuild per EXTRA flag set.
python tools/net_bench.py 2>&1 | tail -3
for X in "$@"; do
  touch rt-octree_amd/csrc/guidance_kernels.hip
  make -C rt-octree_amd/csrc -j8 EXTRA="$X" >/dev/null 2>&1
  echo "EXTRA=$X"; python tools/net_bench.py 2>&1 | tail -3
done
