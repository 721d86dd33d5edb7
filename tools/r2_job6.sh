#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_render_parity.py tests/test_fuzz_parity.py tests/test_quant_direct.py tests/test_expectation_gpu.py -m gpu -q -x > $O/r2_j6_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j6_tests.log
grep -E "passed|failed|rc" $O/r2_j6_tests.log | tail -3
timeout 900 python tools/ab_trav.py 2>&1 | grep -v INFO | tee $O/r2_j6_ab_trav.log
