"""Development harness: time the render kernels on a synthetic tree (GPU box)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=9)
    ap.add_argument("--shell", type=float, default=1.25)
    ap.add_argument("--basis", type=int, default=9)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--spp", type=int, nargs="+", default=[6, 1])
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--kernels", type=int, nargs="+", default=[2, 1])
    ap.add_argument("--strips", type=int, nargs="+", default=[1])
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--tree", default="")
    args = ap.parse_args()
    t0 = time.time()
    if args.tree:
        dt = R.N3Tree(args.tree)
    else:
        tree = synth.make_tree(depth_limit=args.depth, basis_dim=args.basis, shell=args.shell)
        print("tree", tree.stats, "gen %.1fs" % (time.time() - t0), flush=True)
        dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    print("capacity %d device bytes %.1f MB, max_depth %d" % (dt.capacity, dt.device_bytes / 1e6, dt.max_depth))
    W = H = args.size
    poses = synth.orbit_poses(args.frames)
    fx = synth.blender_focal(W)
    cam = R.Camera(W, H, fx, fx)
    ctx = R.RenderContext(W, H)
    ref = {}
    for rnd in range(args.rounds):
        for spp in args.spp:
            opt = R.RenderOptions(spp=spp, denoise=False)
            for k in args.kernels:
                for var in [0]:
                    for strip in (args.strips if k == 2 else [1]):
                        ctx.set_kernel(k)
                        ctx.set_tuning("strip_rows", strip)
                        ctx.rng_seed()
                        cam.set_c2w(poses[0])
                        for _ in range(3):
                            R.launch_renderer(dt, cam, opt, ctx)
                        tm = ctx.timer()
                        tm.reset()
                        times = []
                        for i, p in enumerate(poses):
                            cam.set_c2w(p)
                            ctx.rng_seed()
                            ctx.rng_advance((100 + i) << 32)
                            tm.render_start()
                            R.launch_renderer(dt, cam, opt, ctx)
                            tm.render_stop()
                            tm.record(False)
                        s = tm.stats()
                        aux = ctx.download_aux()
                        key = spp
                        if key not in ref:
                            ref[key] = aux.copy()
                        same = np.array_equal(ref[key].view(np.uint32), aux.view(np.uint32))
                        print("round %d spp %2d kernel %d variant %d strip %d: render %.3f ms/frame (%.0f FPS) same_bits=%s" % (
                            rnd, spp, k, var, strip, s["render_ms"], 1000.0 / s["render_ms"], same), flush=True)


if __name__ == "__main__":
    main()
