#!/bin/bash
OUT=gpurun_out; rm -f $OUT/r3_scratch_hazard_probe.txt
for i in 0 1 2; do
  echo "== $(cat rt-octree_amd/lib_ab/flags_$i.txt)" >> $OUT/r3_scratch_hazard_probe.txt
  RTO_LIB=$PWD/rt-octree_amd/lib_ab/librto_$i.so KINDS="5" bash tools/scratch_hazard_probe.sh 100 > /dev/null
done
echo "== shipped" >> $OUT/r3_scratch_hazard_probe.txt
KINDS="5" bash tools/scratch_hazard_probe.sh 100 > /dev/null
cat $OUT/r3_scratch_hazard_probe.txt
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
PY
timeout 600 python -m pytest tests/test_filter_parity.py -x -q -m gpu 2>&1 | tail -n 1
for i in 1 2; do RTO_LIB=$PWD/rt-octree_amd/lib_ab/librto_$i.so timeout 600 python -m pytest tests/test_filter_parity.py -x -q -m gpu 2>&1 | tail -n 1; done
