#!/bin/bash
# VERDICT r2 task 1: what bounds render_persist?  Run on the GPU box from the repo root: bash tools/calibrate_valu.sh TAG
#  (a) asm VALU probe at 1/2/4/6/8 waves per SIMD, counters on the probe itself (instruction count + clocks)
#  (b) the traversal kernel with its gathers stubbed (rt-octree_amd/lib_ab/librto_1.so built with -DRTO_STUB_LOADS by
#      tools/ab_variants.sh build "" "-DRTO_STUB_LOADS") at 1/2/4/6/8 waves per SIMD: time + SQ_INSTS_VALU / clocks
#  (c) the real kernel: wait / issue / lane-occupancy counters
TAG=${1:-r3}
O=gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python3 tools/probe_valu.py $O/${TAG}_valu_probe.json > $O/${TAG}_valu_probe.txt 2>&1
tail -3 $O/${TAG}_valu_probe.txt
# (a') counters on the probe: kinds 0 (v_fma_f32) and 14 (mix) at 1, 2, 4, 8 waves per SIMD
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $O/${TAG}_pp -- python3 tools/probe_valu.py --kinds 0,14,11 --wps 1,2,4,8 > $O/${TAG}_valu_probe_under_pmc.txt 2>&1
python3 tools/pmc_dump.py $O/${TAG}_valu_probe_pmc.json valu_probe_kernel $O/${TAG}_pp > $O/${TAG}_valu_probe_pmc.txt
rm -rf $O/${TAG}_pp
# (c) the real kernel (C2 raw: the traversal is the same with or without the denoise stage)
B="bench.py --streams 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --steps 2 --warmup 1 --no-denoise"
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACCUM_PREV_HIRES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/${TAG}_real_$i -- python3 $B > /dev/null 2> $O/${TAG}_real_$i.err || tail -2 $O/${TAG}_real_$i.err
done
python3 tools/pmc_summarize.py $O/${TAG}_real_pmc_summary.json $O/${TAG}_real_1 $O/${TAG}_real_2 $O/${TAG}_real_3 $O/${TAG}_real_4 > /dev/null
rm -rf $O/${TAG}_real_[1-4]
# (b) stubbed loads: timing at every occupancy in one process, then one counter pass per occupancy
STUB=$PWD/rt-octree_amd/lib_ab/librto_1.so
if [ -f $STUB ]; then
  RTO_LIB=$STUB python3 tools/ab_tuning.py refill=132 refill=232 refill=432 refill=0 refill=832 > $O/${TAG}_stub_timing.txt 2>&1
  python3 tools/ab_tuning.py refill=132 refill=232 refill=432 refill=0 refill=832 > $O/${TAG}_real_timing_by_occupancy.txt 2>&1
  for T in 132 232 432 0 832; do
    RTO_LIB=$STUB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/${TAG}_stub_$T -- python3 $B --tuning refill=$T > /dev/null 2> $O/${TAG}_stub_$T.err || tail -2 $O/${TAG}_stub_$T.err
    python3 tools/pmc_summarize.py $O/${TAG}_stub_pmc_refill$T.json $O/${TAG}_stub_$T > /dev/null
    rm -rf $O/${TAG}_stub_$T
  done
fi
ls $O | grep ${TAG}_
