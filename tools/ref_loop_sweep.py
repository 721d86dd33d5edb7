"""The reference's loop shape (one operator call + one host wait per frame) through bench.py for several numbers of frames in
flight: python3 tools/ref_loop_sweep.py [inflight ...] [--args "..."]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = []
vals = []
it = iter(sys.argv[1:])
for a in it:
    if a == "--args":
        extra = next(it).split()
    else:
        vals.append(int(a))
for D in vals or [2, 4, 8]:
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-frames", "0", "--psnr-frames", "0",
           "--count-frames", "0", "--spot-pixels", "0", "--no-exact-pass", "--ref-loop-frames", "96", "--ref-loop-inflight", str(D)] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print("inflight %d: failed %s" % (D, p.stderr[-400:]))
        continue
    r = json.loads(lines[-1])["reference_loop"]
    pl = r.get("pipelined") or {}
    print("sequential: fps %.0f (render %.3f net %.3f filter %.3f ms) wall %.0f | in flight %d: wall %.0f fps, host issue %.3f ms + wait %.3f ms per frame, same bits %s"
          % (r["fps"], r["render_ms"], r["torch_ms"], r["filter_ms"], r["wall_fps"], D, pl.get("wall_fps", 0), pl.get("host_issue_ms_per_frame", 0),
             pl.get("host_wait_ms_per_frame", 0), pl.get("last_frame_bit_identical_to_the_sequential_loop")), flush=True)
