#!/bin/bash
# Development: SQ + TCP counter groups of the batched raw render (SPP 6, no denoise), for whatever build is in the tree
O=gpurun_out
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="bench.py --streams 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --steps 2 --warmup 1 --groups-per-step 1 --no-denoise $2"
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/${TAG}_pmc_$i -- python3 $B > /dev/null 2> $O/${TAG}_pmc_$i.err || tail -2 $O/${TAG}_pmc_$i.err
done
python3 tools/pmc_summarize.py $O/${TAG}_pmc_summary.json $O/${TAG}_pmc_1 $O/${TAG}_pmc_2 $O/${TAG}_pmc_3 $O/${TAG}_pmc_4 | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(json.dumps({k:v for k,v in d.items() if k in ('render_persist',)},indent=0))"
rm -rf $O/${TAG}_pmc_[1-4] $O/${TAG}_pmc_[1-4].err
