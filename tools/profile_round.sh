#!/bin/bash
# Evidence run for profiles/: bench lines, kernel trace and separate PMC passes for the BASELINE configurations
# C2 (default), C5 (SPP 1 raw) and C4 (1920x1080 stand-in).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh TAG [c2 c5 c4]   -> gpurun_out/TAG_*
TAG=${1:-rX}; shift
CFGS=${@:-c2 c5 c4}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for C in $CFGS; do
  case $C in
    c2) A="" ; STEPS=2 ;;
    c5) A="--spp 1 --no-denoise" ; STEPS=2 ;;
    c4) A="--c4" ; STEPS=2 ;;
  esac
  B="bench.py $A --streams 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --spot-pixels 0 --count-frames 0 --no-exact-pass --no-full-pass --steps $STEPS --warmup 1 --groups-per-step 1"
  # (the plain bench lines: tools/bench_lines.sh, once the counters of this run are installed)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${C}_trace -- python3 $B > $O/${TAG}_${C}_bench_under_rocprof.json 2>/dev/null
  cp $(find $O/${TAG}_${C}_trace -name '*kernel_stats.csv' | head -1) $O/${TAG}_${C}_kernel_stats.csv
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
             "GRBM_GUI_ACTIVE GRBM_COUNT TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d $O/${TAG}_${C}_pmc_$i -- python3 $B > /dev/null 2> $O/${TAG}_${C}_pmc_$i.err || tail -2 $O/${TAG}_${C}_pmc_$i.err
  done
  python3 tools/pmc_summarize.py $O/${TAG}_${C}_pmc_summary.json $O/${TAG}_${C}_pmc_1 $O/${TAG}_${C}_pmc_2 $O/${TAG}_${C}_pmc_3 $O/${TAG}_${C}_pmc_4 $O/${TAG}_${C}_pmc_5 $O/${TAG}_${C}_pmc_6 > /dev/null
  rm -rf $O/${TAG}_${C}_trace $O/${TAG}_${C}_pmc_[1-6] $O/${TAG}_${C}_pmc_[1-6].err
done
python3 tools/pmc_traffic.py $O/${TAG}_pmc_traffic.json profiles/r2_probe_ceiling.json $(for C in $CFGS; do echo $C=$O/${TAG}_${C}_pmc_summary.json:$O/${TAG}_${C}_bench_under_rocprof.json; done)
