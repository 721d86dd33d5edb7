#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests/test_filter_parity.py tests/test_cli.py tests/test_bench_contract.py -m gpu -q -x > $O/r2_j3_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j3_tests.log
tail -15 $O/r2_j3_tests.log
timeout 300 python tools/filter_bench.py 2>&1 | tee $O/r2_j3_filter_bench.log
timeout 600 python bench.py --cpu-frames 0 > $O/r2_j3_bench.json 2> $O/r2_j3_bench.err; tail -3 $O/r2_j3_bench.err; cut -c1-1500 $O/r2_j3_bench.json
