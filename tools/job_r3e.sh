ROOT=$PWD
O=$ROOT/gpurun_out
cd $ROOT/_ab_old && timeout 300 python3 tools/ab_tuning.py refill=0 2>&1 | grep "round [12]" > $O/r3e_old.txt
cd $ROOT && timeout 600 tools/ab_variants.sh run 2 > $O/r3e_variants.txt 2>&1
timeout 900 python3 -m pytest tests/test_render_parity.py tests/test_fuzz_parity.py tests/test_baseline_workload.py -x -q -m gpu > $O/r3e_pytest.txt 2>&1
tail -3 $O/r3e_pytest.txt; cat $O/r3e_old.txt $O/r3e_variants.txt
