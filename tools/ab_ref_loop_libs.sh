#!/bin/bash
# Same-box A/B of library builds on the reference's loop shape (bench.py's reference_loop: sequential + 4 frames in flight).
# bash tools/ab_ref_loop_libs.sh LIB [LIB ...]   (older revisions' builds from tools/ab_rev.sh load with RTO_LIB_OLDER_BUILD=1)
for r in 1 2; do
for L in "$@"; do
  RTO_LIB=$PWD/$L RTO_LIB_OLDER_BUILD=1 python3 bench.py --streams 1 --steps 2 --warmup 1 --groups-per-step 1 --cpu-frames 0 --psnr-frames 0 --count-frames 0 --spot-pixels 0 --no-exact-pass --no-full-pass --ref-loop-frames 96 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['reference_loop']; p=r['pipelined']
print('round $r %-28s sequential %5.0f fps (render %.3f net %.3f filter %.3f ms)  4 in flight %5.0f wall fps (cull_single %5.0f)  host wait %.3f ms/frame' % ('$(basename $L)', r['fps'], r['render_ms'], r['torch_ms'], r['filter_ms'], p['wall_fps'], p['wall_fps_with_cull_single'], p['host_wait_ms_per_frame']))"
done; done
