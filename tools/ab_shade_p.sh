#!/bin/bash
# Development A/B: pixels per lane (RTO_SHADE_P) of the compacting shade kernel, dense and codebook-direct.
B="python bench.py --streams 1 --steps 96 --warmup 16 --cpu-frames 0 --psnr-frames 0 --no-denoise"
pick() { grep -o '"value": [0-9.]*\|"shade_kernel_avg_launch_ms": [0-9.]*' | tr '\n' ' '; echo; }
$B 2>/dev/null | pick
T=$(ls /dev/shm/rto_bench_tree_*.npz | head -1)
python tools/make_quant_tree.py $T /dev/shm/q.npz --retain 1
$B --tree /dev/shm/q.npz --quant-direct 2>/dev/null | pick
for P in "$@"; do
  touch rt-octree_amd/csrc/render_kernels.hip
  make -C rt-octree_amd/csrc -j8 EXTRA=-DRTO_SHADE_P=$P >/dev/null 2>&1
  echo "P=$P"
  $B 2>/dev/null | pick
  $B --tree /dev/shm/q.npz --quant-direct 2>/dev/null | pick
done
