"""Same-box A/B of builds of librto.so on the reference's loop shape (tools/ref_loop_sweep.py per library, interleaved):
python3 tools/ab_ref_loop.py [--rounds 2] [--inflight 4] LIB [LIB ...]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--inflight", type=int, default=4)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
for r in range(a.rounds):
    for lib in a.libs:
        env = dict(os.environ, RTO_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_loop_sweep.py"), str(a.inflight)], capture_output=True, text=True, env=env, cwd=ROOT)
        print("round %d %-28s %s" % (r, os.path.basename(lib), (p.stdout.strip().splitlines() or [p.stderr[-200:]])[-1]), flush=True)
