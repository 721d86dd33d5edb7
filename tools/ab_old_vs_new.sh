# same-box comparison: the round-2 pipeline (_ab_old: ray set-up inside the persistent loop) vs ray records
ROOT=$PWD
O=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for T in ${@:-old new}; do
  if [ $T = old ]; then D=$ROOT/_ab_old; else D=$ROOT; fi
  cd $D
  timeout 300 python3 tools/ab_tuning.py refill=0 2>&1 | grep "round [12]" > $O/r3d_${T}_timing.txt
  RTO_LIB=$D/rt-octree_amd/lib_dbg/librto.so timeout 300 python3 tools/dbg_counters.py 2>&1 | grep refill > $O/r3d_${T}_dbg.txt
  B="bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --steps 2 --warmup 1 --no-denoise"
  i=0
  for SET in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
             "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $SET --output-format csv -d $O/r3d_${T}_$i -- python3 $B > /dev/null 2> $O/r3d_${T}_$i.err || tail -2 $O/r3d_${T}_$i.err
  done
  python3 tools/pmc_summarize.py $O/r3d_${T}_pmc.json $O/r3d_${T}_1 $O/r3d_${T}_2 $O/r3d_${T}_3 $O/r3d_${T}_4 > /dev/null
  rm -rf $O/r3d_${T}_[1-4] $O/r3d_${T}_[1-4].err
done
cat $O/r3d_*_timing.txt $O/r3d_*_dbg.txt
