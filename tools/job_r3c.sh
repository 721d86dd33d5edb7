O=gpurun_out
timeout 1500 python3 -m pytest tests/test_render_parity.py tests/test_fuzz_parity.py tests/test_quant_direct.py tests/test_big_tree.py -x -q -m gpu > $O/r3c_pytest.txt 2>&1
tail -5 $O/r3c_pytest.txt
python3 tools/ab_tuning.py refill=0 refill=616 refill=608 refill=832 refill=816 refill=808 refill=804 blocks_per_cu=6,refill=816 blocks_per_cu=7,refill=816 > $O/r3c_ab.txt 2>&1
grep "round [12]" $O/r3c_ab.txt
