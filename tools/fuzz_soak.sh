#!/bin/bash
# Soak run of the random-scene parity test (every traversal kernel against the oracle, bit for bit) and of the random-frame
# test of the culled denoise stage on the GPU box:
# bash tools/fuzz_soak.sh FIRST LAST [workers]  -> gpurun_out/fuzz_soak_FIRST_LAST.txt
A=${1:-32}; B=${2:-532}; N=${3:-8}
O=gpurun_out; mkdir -p $O
RTO_FUZZ_SEEDS=$A:$B OMP_NUM_THREADS=16 timeout 2400 python3 -m pytest tests/test_fuzz_parity.py tests/test_filter_cull.py::test_random_frames_culled_denoise_is_bit_identical -q -m gpu -n $N -p no:cacheprovider 2>&1 | grep -v "^INFO" | tail -n 15 > $O/fuzz_soak_${A}_${B}.txt
tail -n 5 $O/fuzz_soak_${A}_${B}.txt
