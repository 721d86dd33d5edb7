#!/bin/bash
# debug build with the per-branch counters of the traversal loop: -> rt-octree_amd/lib_dbg/librto.so  [extra flags]
cd "$(dirname "$0")/.."
make -C rt-octree_amd/csrc -j8 OUT=../lib_dbg OBJ=../lib_dbg/obj EXTRA="-DRTO_DBG_COUNTERS $*" ../lib_dbg/librto.so 2>&1 | grep -i " error\|warning: unused" ; ls -la rt-octree_amd/lib_dbg/librto.so
