#!/bin/bash
# Development A/B on one box: bench the in-tree build, then rebuilds with each EXTRA flag set given
# as an argument (e.g. tools/ab_build.sh -DRTO_SETUP_HOIST "-DFOO=2"), two runs each.
B="python bench.py --streams 1 --steps 4 --warmup 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-denoise"
pick() { grep -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"shade_kernel_avg_launch_ms": [0-9.]*' | tr '\n' ' '; echo; }
echo "default"; $B 2>/dev/null | pick; $B 2>/dev/null | pick
for X in "$@"; do
  touch rt-octree_amd/csrc/render_kernels.hip
  make -C rt-octree_amd/csrc -j8 EXTRA="$X" >/dev/null 2>&1
  echo "EXTRA=$X"; $B 2>/dev/null | pick; $B 2>/dev/null | pick
done
touch rt-octree_amd/csrc/render_kernels.hip
make -C rt-octree_amd/csrc -j8 >/dev/null 2>&1
echo "default again"; $B 2>/dev/null | pick
