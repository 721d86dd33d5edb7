#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1800 python -m pytest tests -m gpu -q > $O/r2_j11_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j11_tests.log
grep -E "passed|failed|rc|FAILED|Error" $O/r2_j11_tests.log | tail -6
bash tools/profile_round.sh r2_c c2 c5 c4 > $O/r2_c_profile.log 2>&1; tail -3 $O/r2_c_profile.log
for f in $O/r2_c_bench_c2.json $O/r2_c_bench_c5.json $O/r2_c_bench_c4.json; do grep '^{' $f | cut -c1-200; done
