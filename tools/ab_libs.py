"""Same-box A/B of builds of librto.so through bench.py itself: each library in turn (RTO_LIB), interleaved rounds, the
per-kernel launch durations and frames/s of the bench workload.  python3 tools/ab_libs.py [--rounds 3] [--args "--c4"] LIB [LIB ...]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--args", default="")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-frames", "0", "--psnr-frames", "0", "--ref-loop-frames", "0", "--streams", "1",
            "--no-exact-pass", "--no-full-pass", "--count-frames", "0", "--spot-pixels", "16", "--steps", "6", "--warmup", "2", "--groups-per-step", "1"] + a.args.split()
    for r in range(a.rounds):
        for lib in a.libs:
            env = dict(os.environ, RTO_LIB=os.path.abspath(lib))
            p = subprocess.run(base, capture_output=True, text=True, env=env, cwd=ROOT)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode or not lines:
                print("round %d %s: FAILED rc %d %s" % (r, lib, p.returncode, p.stderr[-300:]), flush=True)
                continue
            d = json.loads(lines[-1])
            rf, rt = d["roofline"], d["reference_timer"]
            print("round %d %-44s %8.0f frames/s  marks+lists+thresholds %.3f  traverse %.3f  shade %.3f  net %.4f  filter %.4f ms/frame x100  "
                  "parity mismatches %s" % (r, os.path.basename(lib), d["value"], rf["thresholds_kernel_avg_launch_ms"], rf["avg_launch_ms"],
                                           rf["shade_kernel_avg_launch_ms"], rt["torch_ms"] * 100, rt["filter_ms"] * 100,
                                           (d.get("parity_spot") or {}).get("mismatches")), flush=True)


if __name__ == "__main__":
    main()
