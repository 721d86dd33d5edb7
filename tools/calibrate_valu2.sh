#!/bin/bash
# Second calibration pass (VERDICT r2 task 1): per-opcode VALU issue rates with counters on the probe itself, and the
# traversal kernel -- real and with stubbed gathers -- at a TRUE occupancy of 1..6 workgroups per CU (tuning key
# blocks_per_cu: the persistent grid is CUs x k workgroups, i.e. k waves per SIMD).
TAG=${1:-r3b}
O=gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python3 tools/probe_valu.py $O/${TAG}_valu_probe.json --wps 1,2,4,8 > $O/${TAG}_valu_probe.txt 2>&1
tail -2 $O/${TAG}_valu_probe.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_pp -- python3 tools/probe_valu.py --wps 2,8 > $O/${TAG}_valu_probe_under_pmc.txt 2>&1
python3 tools/pmc_dump.py $O/${TAG}_valu_probe_pmc.json valu_probe_kernel $O/${TAG}_pp > /dev/null
rm -rf $O/${TAG}_pp
STUB=$PWD/rt-octree_amd/lib_ab/librto_1.so
S="blocks_per_cu=1 blocks_per_cu=2 blocks_per_cu=3 blocks_per_cu=4 blocks_per_cu=5 blocks_per_cu=6"
python3 tools/ab_tuning.py $S > $O/${TAG}_real_by_occupancy.txt 2>&1
RTO_LIB=$STUB python3 tools/ab_tuning.py $S > $O/${TAG}_stub_by_occupancy.txt 2>&1
B="bench.py --streams 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --steps 2 --warmup 1 --no-denoise"
for K in 1 2 4 6; do
  RTO_LIB=$STUB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_stub_$K -- python3 $B --tuning blocks_per_cu=$K > /dev/null 2> $O/${TAG}_stub_$K.err || tail -2 $O/${TAG}_stub_$K.err
  python3 tools/pmc_summarize.py $O/${TAG}_stub_pmc_k$K.json $O/${TAG}_stub_$K > /dev/null
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_real_$K -- python3 $B --tuning blocks_per_cu=$K > /dev/null 2> $O/${TAG}_real_$K.err || tail -2 $O/${TAG}_real_$K.err
  python3 tools/pmc_summarize.py $O/${TAG}_real_pmc_k$K.json $O/${TAG}_real_$K > /dev/null
  rm -rf $O/${TAG}_stub_$K $O/${TAG}_real_$K $O/${TAG}_stub_$K.err $O/${TAG}_real_$K.err
done
timeout 1500 python3 -m pytest tests/test_baseline_workload.py tests/test_render_parity.py tests/test_guidance_fused.py tests/test_abi_and_host.py -x -q -m gpu > $O/${TAG}_pytest.txt 2>&1
tail -5 $O/${TAG}_pytest.txt
