#!/bin/bash
# Evidence run for profiles/: kernel trace + separate PMC passes of the default bench workload.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh TAG   -> gpurun_out/TAG_*
TAG=${1:-rX}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="bench.py --cpu-frames 0 --psnr-frames 0 --steps 96 --warmup 32"
python3 bench.py --cpu-frames 2 > $O/${TAG}_bench_c2.json 2> $O/${TAG}_bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $B > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
cp $(find $O/${TAG}_trace -name '*kernel_stats.csv' | head -1) $O/${TAG}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $B > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/${TAG}_pmc_tcc -- python3 $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/${TAG}_pmc_sq -- python3 $B > /dev/null 2>&1
python3 tools/pmc_summarize.py $O/${TAG}_pmc_summary.json $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_tcc $O/${TAG}_pmc_sq --traffic $O/${TAG}_pmc_traffic.json
rm -rf $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_tcc $O/${TAG}_pmc_sq
# the other configurations of BASELINE.json: C5 (SPP 1 raw) and the C4 shape (1920x1080, SH25, denoise)
python3 bench.py --spp 1 --no-denoise --cpu-frames 0 --psnr-frames 0 --steps 192 --warmup 32 > $O/${TAG}_bench_c5_spp1_raw.json 2>/dev/null
python3 bench.py --width 1920 --height 1080 --basis 25 --depth 9 --cpu-frames 0 --psnr-frames 0 --steps 96 --warmup 32 > $O/${TAG}_bench_c4_1080p_sh25.json 2>/dev/null
