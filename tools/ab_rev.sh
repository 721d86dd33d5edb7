#!/bin/bash
# Build librto.so of another git revision next to the working tree's, for a same-box A/B (boxes differ by +-5 %):
#   here (no GPU):  tools/ab_rev.sh <rev> <name>      -> rt-octree_amd/lib_ab/librto_<name>.so   (csrc of <rev>, this Makefile)
#   on the box:     python3 tools/ab_libs.py [--args "..."] rt-octree_amd/lib/librto.so rt-octree_amd/lib_ab/librto_<name>.so
set -e
cd "$(dirname "$0")/.."
REV=$1; NAME=$2; T=$(mktemp -d /tmp/rto_ab_XXXX)
mkdir -p $T/repo/rt-octree_amd rt-octree_amd/lib_ab
git archive $REV rt-octree_amd/csrc include | tar -x -C $T/repo
make -C $T/repo/rt-octree_amd/csrc -j8 ../lib/librto.so > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
cp $T/repo/rt-octree_amd/lib/librto.so rt-octree_amd/lib_ab/librto_$NAME.so
echo "$REV" > rt-octree_amd/lib_ab/flags_$NAME.txt
rm -rf $T; ls -la rt-octree_amd/lib_ab/librto_$NAME.so
