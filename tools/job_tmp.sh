#!/bin/bash
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
: > $OUT/soak_failures.txt
for rep in 1 2 3 4 5 6 7 8 9 10; do
RTO_FUZZ_SEEDS=0:3000 OMP_NUM_THREADS=16 timeout 1200 python3 -m pytest tests/test_filter_cull.py::test_random_frames_culled_denoise_is_bit_identical tests/test_fuzz_parity.py -q -m gpu -n 8 -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|^E  " | cut -c1-600 | tail -n 12 >> $OUT/soak_failures.txt
done
grep -c "passed" $OUT/soak_failures.txt; grep -v "^6000 passed" $OUT/soak_failures.txt | head -30
