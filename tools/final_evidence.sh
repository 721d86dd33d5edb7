#!/bin/bash
# Round-3 evidence bundle (run on the GPU box from the repo root): bash tools/final_evidence.sh [tag] -> gpurun_out/<tag>_*
T=${1:-r3_k}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 bash tools/profile_round.sh ${T} c2 c5 c4 > $O/${T}_profile.log 2>&1; tail -2 $O/${T}_profile.log
timeout 900 python3 bench.py --scenes 8 --scene-map both --steps 16 --warmup 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass > $O/${T}_bench_c3_8scenes_1gpu.json 2> $O/${T}_bench_c3.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_cmd.json 2> $O/${T}_bench_driver_cmd.err
# the traversal loop with its gathers stubbed, current code: the VALU ceiling of ITS instruction stream
STUB=$PWD/rt-octree_amd/lib_ab/librto_1.so
if [ -f $STUB ]; then
  RTO_LIB=$STUB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${T}_stub -- python3 bench.py --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --spot-pixels 0 --no-exact-pass --steps 2 --warmup 1 --no-denoise > /dev/null 2> $O/${T}_stub.err
  python3 tools/pmc_summarize.py $O/${T}_stubbed_loads_pmc.json $O/${T}_stub > /dev/null; rm -rf $O/${T}_stub
fi
for f in $O/${T}_bench_c2.json $O/${T}_bench_c5.json $O/${T}_bench_c4.json $O/${T}_bench_c3_8scenes_1gpu.json $O/${T}_bench_driver_cmd.json; do grep '^{' $f | cut -c1-170; done
