#!/bin/bash
# Development: L1 counters of the traversal kernel (separate passes, 4 counters each).
# (No TA_* / TD_* sets: on this pool a pass with TA_TA_BUSY_sum / TA_ADDR_STALLED_BY_TC_CYCLES_sum aborts inside rocprofv3 and
#  then hangs in its finalisation until the job's limit -- round 3 lost 65 GPU-minutes to it.  Every pass runs under `timeout`.)
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="bench.py --streams 1 --cpu-frames 0 --psnr-frames 0 --steps 32 --warmup 16 --no-denoise"
i=0
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/px_$i -- python3 $B > /dev/null 2> $O/px_$i.err || tail -3 $O/px_$i.err
done
python3 tools/pmc_summarize.py $O/pmc_extra.json $O/px_1 $O/px_2 $O/px_3 $O/px_4 | grep -A24 render_persist
rm -rf $O/px_*
