#!/bin/bash
OUT=gpurun_out; rm -f $OUT/r3_scratch_hazard_probe.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 1
echo "--- processes right after pytest:"; ps -eo pid,etime,pcpu,cmd | grep -E "python|volrend|torch" | grep -v grep | cut -c1-150 | head
rocm-smi --showuse --showpower --showtemp --showclocks 2>/dev/null | grep -E "GPU\[0\]" | head -12
KINDS="-1 -1" bash tools/scratch_hazard_probe.sh 100 > /dev/null
rocm-smi --showuse --showpower --showtemp 2>/dev/null | grep -E "GPU\[0\]" | head -8
cat $OUT/r3_scratch_hazard_probe.txt
