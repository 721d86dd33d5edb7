#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1800 python -m pytest tests -m gpu -q > $O/r2_j7_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j7_tests.log
grep -E "passed|failed|rc|FAILED|Error" $O/r2_j7_tests.log | tail -8
bash tools/profile_round.sh r2_b c2 c5 c4 > $O/r2_b_profile.log 2>&1; tail -5 $O/r2_b_profile.log
for f in $O/r2_b_bench_c2.json $O/r2_b_bench_c5.json $O/r2_b_bench_c4.json; do grep '^{' $f | cut -c1-300; done
