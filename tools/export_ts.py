"""Exports a trained GuidanceNet state_dict (default: the committed rt-octree_amd/weights/guidance_synth_lego.pt)
as the TorchScript module volrend_headless loads with --ts_module -- the reference's own artefact
(denoiser/network.py:170-208 compact_and_compile: fold the branches, cast to fp16, jit.trace).
usage: python tools/export_ts.py [weights.pt] [out.ts]   (needs a HIP device: the fp16 module is traced on it)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rt_octree_amd import denoiser  # noqa: E402


def main():
    wpath = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rt-octree_amd", "weights", "guidance_synth_lego.pt")
    out = sys.argv[2] if len(sys.argv) > 2 else "ts_latest.ts"
    model = denoiser.GuidanceNet(8, 32, 5, 2, 4)
    model.load_state_dict(torch.load(wpath, map_location="cpu"))
    dev = "cuda:0" if torch.cuda.is_available() else None
    ts = denoiser.compact_and_compile(model, device=dev, example_hw=(800, 800))
    ts.save(out)
    print("wrote", out, "(fp16, traced on %s)" % dev if dev else "(fp32, CPU)")


if __name__ == "__main__":
    main()
