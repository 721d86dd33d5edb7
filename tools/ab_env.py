"""Same-box A/B of ENVIRONMENT settings (upload-time switches such as RTO_TOP_LEVELS) through bench.py: interleaved rounds, per-kernel
launch durations and frames/s.  python3 tools/ab_env.py [--rounds 2] [--args "--c4"] "" RTO_TOP_LEVELS=7 "RTO_TOP_LEVELS=8 FOO=1" """
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--args", default="")
    ap.add_argument("envs", nargs="+")
    a = ap.parse_args()
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-frames", "0", "--psnr-frames", "0", "--ref-loop-frames", "0", "--streams", "1",
            "--no-exact-pass", "--no-full-pass", "--count-frames", "0", "--spot-pixels", "16", "--steps", "6", "--warmup", "2", "--groups-per-step", "1"] + a.args.split()
    for r in range(a.rounds):
        for ev in a.envs:
            env = dict(os.environ)
            for kv in ev.split():
                k, v = kv.split("=", 1)
                env[k] = v
            p = subprocess.run(base, capture_output=True, text=True, env=env, cwd=ROOT)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode or not lines:
                print("round %d [%s]: FAILED rc %d %s" % (r, ev, p.returncode, p.stderr[-300:]), flush=True)
                continue
            d = json.loads(lines[-1])
            rf, rt = d["roofline"], d["reference_timer"]
            print("round %d %-28s %8.0f frames/s  marks+lists+thresholds %.3f  traverse %.3f  shade %.3f  net %.4f  filter %.4f ms/frame x100  "
                  "tree %.0f MB  parity mismatches %s" % (r, "[" + ev + "]", d["value"], rf["thresholds_kernel_avg_launch_ms"], rf["avg_launch_ms"],
                                                          rf["shade_kernel_avg_launch_ms"], rt["torch_ms"] * 100, rt["filter_ms"] * 100,
                                                          d["config"]["tree_device_mb"], (d.get("parity_spot") or {}).get("mismatches")), flush=True)


if __name__ == "__main__":
    main()
