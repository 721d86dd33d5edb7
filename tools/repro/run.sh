#!/bin/bash
# On the GPU box: build the stand-alone reproducer and run its victim alone, then beside 7 MFMA "other processes".
# bash tools/repro/run.sh [seconds] -> gpurun_out/r3_pkfma_repro.txt
set -e
S=${1:-20}; O=gpurun_out; mkdir -p $O
B=/tmp/pkfma_repro
hipcc --offload-arch=gfx950 -O2 tools/repro/pkfma_repro.hip -o $B 2> $O/pkfma_build.err
/opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading $B > /dev/null 2>&1 || true
{
echo "== victim alone on the GPU ($S s each)"
$B check $S pk
$B check $S scalar
for V in pk scalar; do
  echo "== victim ($V) beside 7 processes running dependent MFMA chains"
  for i in 1 2 3 4 5 6 7; do $B load $((S + 4)) > /dev/null & done
  sleep 2
  $B check $S $V
  wait
done
} | tee $O/r3_pkfma_repro.txt
