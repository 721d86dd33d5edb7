ROOT=$PWD
O=$ROOT/gpurun_out
timeout 3000 bash tools/profile_round.sh r3_g c2 c5 c4
for D in 2 3 6 8; do
  timeout 300 python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --psnr-frames 0 --spot-pixels 0 --no-exact-pass --ref-loop-frames 96 --ref-loop-inflight $D 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
rl=d['reference_loop']; print('inflight', rl['pipelined']['frames_in_flight'], 'wall_fps %.0f'%rl['pipelined']['wall_fps'], 'seq fps %.0f wall %.0f'%(rl['fps'], rl['wall_fps']), rl['pipelined']['last_frame_bit_identical_to_the_sequential_loop'])
" >> $O/r3_g_inflight.txt
done
cat $O/r3_g_inflight.txt
timeout 900 python3 -m pytest tests/test_cli.py tests/test_bench_contract.py tests/test_sharding.py -x -q -m gpu > $O/r3_g_pytest.txt 2>&1; tail -3 $O/r3_g_pytest.txt
