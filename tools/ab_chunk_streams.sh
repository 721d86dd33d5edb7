for r in 1 2 3; do for T in 4832 2832; do python3 bench.py --tuning refill=$T --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass --no-full-pass --count-frames 0 --spot-pixels 16 --steps 8 --warmup 2 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print('round $r refill=$T two-stream value %.0f single %.0f' % (d['value'], d['value_single_stream']))
"; done; done
