#!/bin/bash
# round-2 job 1: baseline tests, ceiling probes, ordered vs shuffled tree
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests -m gpu -x -q > $O/r2_j1_tests.log 2>&1; echo "tests rc $?" >> $O/r2_j1_tests.log
timeout 900 python tools/probe_ceiling.py $O/r2_probe_ceiling.json > $O/r2_probe_ceiling.log 2>&1
timeout 300 python bench.py --cpu-frames 0 --psnr-frames 0 > $O/r2_j1_bench_ordered.json 2> $O/r2_j1_bench_ordered.err
timeout 400 python bench.py --cpu-frames 0 --psnr-frames 0 --shuffle-nodes 1 > $O/r2_j1_bench_shuffled.json 2> $O/r2_j1_bench_shuffled.err
tail -3 $O/r2_j1_tests.log; tail -5 $O/r2_probe_ceiling.log; cat $O/r2_j1_bench_ordered.json | cut -c1-400; cat $O/r2_j1_bench_shuffled.json | cut -c1-400
