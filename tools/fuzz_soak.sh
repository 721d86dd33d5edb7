#!/bin/bash
# Soak run of the random-scene parity test (every traversal kernel against the oracle, bit for bit) and of the random-frame
# test of the culled denoise stage on the GPU box; a failure is reported WITH ITS SEED (the pytest id of the case) and the
# first differing word, so it can be re-run alone: RTO_FUZZ_SEEDS=<seed>:<seed+1> python -m pytest tests/test_fuzz_parity.py -m gpu
# bash tools/fuzz_soak.sh FIRST LAST [workers]  -> gpurun_out/fuzz_soak_FIRST_LAST.txt
A=${1:-32}; B=${2:-532}; N=${3:-8}
O=gpurun_out; mkdir -p $O
RTO_FUZZ_SEEDS=$A:$B OMP_NUM_THREADS=16 timeout ${RTO_SOAK_TIMEOUT:-2400} python3 -m pytest tests/test_fuzz_parity.py tests/test_filter_cull.py::test_random_frames_culled_denoise_is_bit_identical -q -m gpu -n $N -rf -p no:cacheprovider > $O/fuzz_soak_${A}_${B}.full.txt 2>&1
{ grep -E "^(FAILED|ERROR)|AssertionError|words differ" $O/fuzz_soak_${A}_${B}.full.txt | head -n 40; grep -v "^INFO" $O/fuzz_soak_${A}_${B}.full.txt | tail -n 5; } > $O/fuzz_soak_${A}_${B}.txt
rm -f $O/fuzz_soak_${A}_${B}.full.txt
cat $O/fuzz_soak_${A}_${B}.txt
