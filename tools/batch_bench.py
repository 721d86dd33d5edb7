"""Development harness: one-frame launches (render_fast) vs batched launches (render_persist)."""
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
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--shell", type=float, default=2.5)
    ap.add_argument("--basis", type=int, default=16)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--spp", type=int, default=6)
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--refills", type=int, nargs="+", default=[0])
    ap.add_argument("--tree", default="")
    args = ap.parse_args()
    import torch
    if args.tree:
        dt = R.N3Tree(args.tree)
    else:
        t0 = time.time()
        tree = synth.make_tree(depth_limit=args.depth, basis_dim=args.basis, shell=args.shell)
        print("tree", tree.stats, "gen %.1fs" % (time.time() - t0), flush=True)
        dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    W = H = args.size
    poses = synth.orbit_poses(200)[:args.frames]
    fx = synth.blender_focal(W)
    cams = []
    for p in poses:
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(p)
        cams.append(c)
    opt = R.RenderOptions(spp=args.spp, denoise=False)
    ctx = R.RenderContext(W, H, frames=32)
    stream = torch.cuda.current_stream()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / len(cams)

    def single():
        ctx.select_frame(0)
        for i, c in enumerate(cams):
            ctx.rng_seed()
            ctx.rng_advance((100 + i) << 32)
            R.launch_renderer(dt, c, opt, ctx, stream)

    ctx.set_kernel(R.KERNEL_FAST)
    ref = None
    for rnd in range(args.rounds):
        print("round %d: one frame per launch (render_fast): %.3f ms/frame" % (rnd, timed(single)), flush=True)
        if ref is None:
            ref = ctx.download_aux()  # last frame
        for B, RF in [(b, r) for b in args.batches for r in args.refills]:
            ctx.set_tuning("refill", abs(RF))
            ctx.set_tuning("tile_order", 0 if RF < 0 else 1)  # negative refill = row-major tile order

            def batched():
                for i in range(0, len(cams), B):
                    grp = cams[i:i + B]
                    ctx.rng_seed()
                    R.launch_renderer_batch(dt, grp, opt, ctx, stream, rng_jumps=[100 + i + k for k in range(len(grp))])
            ms = timed(batched)
            last = (len(cams) - 1) % B
            ctx.select_frame(last)
            same = np.array_equal(ctx.download_aux().view(np.uint32), ref.view(np.uint32))
            ctx.select_frame(0)
            print("round %d: batch %d refill %d (render_persist): %.3f ms/frame  same_bits=%s" % (rnd, B, RF, ms, same), flush=True)


if __name__ == "__main__":
    main()
