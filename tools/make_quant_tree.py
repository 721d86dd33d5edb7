"""Development helper: quantise a dense synthetic tree.npz (synth.SynthTree.save_quant_npz) so the
codebook-direct render path can be benchmarked against the expanded one on the same scene."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rt_octree_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dense")
    ap.add_argument("out")
    ap.add_argument("--retain", type=int, default=1)
    ap.add_argument("--quantiser", choices=("luminance", "median_cut"), default="luminance")
    a = ap.parse_args()
    z = np.load(a.dense)
    t = synth.SynthTree(z["child"], z["data"], z["invradius3"], z["offset"], str(z["data_format"]), 0, {})
    t.save_quant_npz(a.out, n_retain=a.retain, quantiser=a.quantiser)
    print("dense %.1f MB -> quantised %.1f MB" % (os.path.getsize(a.dense) / 1e6, os.path.getsize(a.out) / 1e6))


if __name__ == "__main__":
    main()
