#!/bin/bash
# On the GPU box: build the stand-alone reproducer and run its victim alone, then beside 7 MFMA "other processes".
# bash tools/repro/run.sh [seconds] -> gpurun_out/r3_pkfma_repro.txt
set -e
S=${1:-20}; O=gpurun_out; mkdir -p $O
B=/tmp/pkfma_repro
hipcc --offload-arch=gfx950 -O2 tools/repro/pkfma_repro.hip -o $B 2> $O/pkfma_build.err
/opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading $B > /dev/null 2>&1 || true
{
if [ -z "$FILTER_ONLY" ]; then
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
fi
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -Irt-octree_amd/csrc"
hipcc $F -DRTO_FILTER_PK=1 tools/repro/filter_repro.hip -o /tmp/filter_repro_pk 2>> $O/pkfma_build.err
hipcc $F -fno-slp-vectorize -DRTO_FILTER_PK=0 tools/repro/filter_repro.hip -o /tmp/filter_repro_scalar 2>> $O/pkfma_build.err
for V in pk scalar; do
  echo "== the library's bit-exact filter kernel ($V build) on pseudo-random inputs, alone, then beside 7 MFMA processes"
  /tmp/filter_repro_$V 8
  for i in 1 2 3 4 5 6 7; do $B load $((S + 4)) > /dev/null & done
  sleep 2
  /tmp/filter_repro_$V $S
  wait
  echo "== ... ONE process launching the MFMA + scratch kernel itself before every filter launch, alone on the GPU"
  /tmp/filter_repro_$V 8 own_mfma
  echo "== ... 8 processes, each launching the MFMA + scratch kernel itself before every filter launch (the library harness's shape)"
  for i in 1 2 3 4 5 6 7; do /tmp/filter_repro_$V $((S + 2)) own_mfma > /dev/null & done
  sleep 1
  /tmp/filter_repro_$V $S own_mfma
  wait
done
} | tee $O/r3_pkfma_repro${FILTER_ONLY:+_filter}.txt
