#!/bin/bash
# Evidence bundle of a round (run on the GPU box from the repo root): bash tools/final_evidence.sh [tag] -> gpurun_out/<tag>_*
# Order matters: (1) the traversal loop with its gathers stubbed -> the VALU ceiling of THIS code's instruction stream,
# (2) kernel trace + counter passes of the real kernels (tools/profile_round.sh) -> <tag>_pmc_traffic.json, installed as
# profiles/pmc_traffic.json in this copy of the tree, (3) the plain bench lines, which then carry those counters
# (a line run before (2) says traffic_stale).  Copy gpurun_out/<tag>_* into profiles/ afterwards.
T=${1:-r4_z}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
STUB=$PWD/rt-octree_amd/lib_ab/librto_1.so   # tools/ab_variants.sh build "" "-DRTO_STUB_LOADS"
if [ -f $STUB ]; then
  RTO_LIB=$STUB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${T}_stub -- python3 bench.py --streams 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --spot-pixels 0 --count-frames 0 --no-exact-pass --no-full-pass --steps 2 --warmup 1 --groups-per-step 1 --no-denoise > /dev/null 2> $O/${T}_stub.err
  python3 tools/pmc_summarize.py $O/${T}_stubbed_loads_pmc.json $O/${T}_stub > /dev/null; rm -rf $O/${T}_stub
  cp $O/${T}_stubbed_loads_pmc.json profiles/
fi
timeout 2400 bash tools/profile_round.sh ${T} c2 c5 c4 > $O/${T}_profile.log 2>&1; tail -2 $O/${T}_profile.log
cp $O/${T}_pmc_traffic.json profiles/pmc_traffic.json
bash tools/bench_lines.sh ${T}
