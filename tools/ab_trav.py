"""Development A/B on one box: batched traversal with the per-slot image (trav=0) vs the header image (trav=1),
on the bench tree as generated (breadth-first) and with its nodes shuffled on disk order."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import synth  # noqa: E402


def main():
    import torch
    base = synth.make_tree(depth_limit=10, basis_dim=16, shell=2.5)
    W = H = 800
    fx = synth.blender_focal(W)
    cams = []
    for p in synth.orbit_poses(200)[:96]:
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(p)
        cams.append(c)
    opt = R.RenderOptions(spp=6, denoise=False)
    B = 32
    stream = torch.cuda.current_stream()
    ref = None
    for name, tree in (("ordered", base), ("shuffled", synth.shuffle_nodes(base, 1))):
        dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
        ctx = R.RenderContext(W, H, frames=B)
        for rnd in range(2):
            for trav in (0, 1):
                ctx.set_tuning("trav", trav)
                ctx.kernel_timing(True)
                for rep in range(2):
                    for i in range(0, len(cams), B):
                        ctx.rng_seed()
                        R.launch_renderer_batch(dt, cams[i:i + B], opt, ctx, stream, rng_jumps=[100 + i + k for k in range(B)])
                    torch.cuda.synchronize()
                    kt = ctx.kernel_timing_read()
                ctx.select_frame(B - 1)
                aux = ctx.download_aux()
                if ref is None:
                    ref = aux
                same = np.array_equal(aux.view(np.uint32), ref.view(np.uint32))
                print("%-9s round %d trav %d: traverse %.3f ms  shade %.3f ms per launch  same_bits=%s"
                      % (name, rnd, trav, kt["traverse_ms"], kt["shade_ms"], same), flush=True)
        dt.free()


if __name__ == "__main__":
    main()
