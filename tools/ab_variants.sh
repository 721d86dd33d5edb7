#!/bin/bash
# Same-box A/B of BUILDS of librto.so (box-to-box differences of the VALU-bound traversal kernel reach 10 %, so two
# builds are only comparable inside one gpurun call).
#   here (no GPU):   tools/ab_variants.sh build  "" "-DFOO" "-DBAR=2"     -> rt-octree_amd/lib_ab/librto_<i>.so
#   on the GPU box:  tools/ab_variants.sh run [rounds]                     -> traversal ms per variant, interleaved rounds
cd "$(dirname "$0")/.."
D=rt-octree_amd/lib_ab
if [ "$1" = build ]; then
  shift; rm -rf $D; mkdir -p $D; i=0
  for X in "$@"; do
    touch rt-octree_amd/csrc/*.hip rt-octree_amd/csrc/*.cpp
    make -C rt-octree_amd/csrc -j8 EXTRA="$X" >/dev/null 2>&1 || { echo "build failed: $X"; exit 1; }
    cp rt-octree_amd/lib/librto.so $D/librto_$i.so; echo "$X" > $D/flags_$i.txt; i=$((i+1))
  done
  touch rt-octree_amd/csrc/*.hip rt-octree_amd/csrc/*.cpp; make -C rt-octree_amd/csrc -j8 >/dev/null 2>&1
  ls -la $D
else
  R=${2:-3}
  for r in $(seq 1 $R); do
    for f in $D/librto_*.so; do
      i=${f##*_}; i=${i%.so}
      echo "round $r variant $i [$(cat $D/flags_$i.txt)]: $(RTO_LIB=$PWD/$f python tools/ab_tuning.py tile_major=1 2>&1 | grep 'round 2' | sed 's/.*traverse/traverse/')"
    done
  done
fi
