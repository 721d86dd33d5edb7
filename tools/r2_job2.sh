#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests -m gpu -q > $O/r2_j2_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j2_tests.log
tail -5 $O/r2_j2_tests.log
