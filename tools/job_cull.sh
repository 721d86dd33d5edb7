ROOT=$PWD
O=$ROOT/gpurun_out
timeout 900 python3 -m pytest tests/test_culling.py tests/test_render_parity.py tests/test_fuzz_parity.py tests/test_quant_direct.py tests/test_big_tree.py tests/test_baseline_workload.py tests/test_c4_full_size.py tests/test_expectation_gpu.py -x -q -m gpu > $O/r3_cull_pytest.txt 2>&1
grep -E "passed|failed" $O/r3_cull_pytest.txt | tail -2
timeout 600 python3 tools/ab_tuning.py cull=1 cull=0 2>&1 | grep "round [12]" > $O/r3_cull_ab.txt; cat $O/r3_cull_ab.txt
