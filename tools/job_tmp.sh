#!/bin/bash
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^INFO: Scale" > $OUT/r3_l_pytest_gpu.txt
tail -n 3 $OUT/r3_l_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^INFO" | tail -n 2
