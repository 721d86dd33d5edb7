"""Development A/B of rto_ctx_set_tuning keys on the batched path (same box, same process):
python tools/ab_tuning.py xcd_queues=0 xcd_queues=1 ...   -> ms per 16-frame launch, traversal kernel."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import synth  # noqa: E402


def main():
    import torch
    settings = [a for a in sys.argv[1:] if "=" in a] or ["xcd_queues=0", "xcd_queues=1"]
    cache = "/dev/shm/rto_ab_tree_d10_b16.npz"  # (20 s to generate: shared by the runs of one A/B job)
    if os.path.exists(cache):
        z = np.load(cache)
        dt = R.N3Tree.from_arrays(z["child"], z["data"], z["scale"], z["offset"], str(z["data_format"]))
    else:
        tree = synth.make_tree(depth_limit=10, basis_dim=16, shell=2.5)
        np.savez(cache + ".tmp.npz", child=tree.child, data=tree.data, scale=tree.scale, offset=tree.offset,
                 data_format=tree.data_format)
        os.replace(cache + ".tmp.npz", cache)
        dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    W = H = 800
    fx = synth.blender_focal(W)
    cams = []
    for p in synth.orbit_poses(200)[:200]:
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(p)
        cams.append(c)
    opt = R.RenderOptions(spp=6, denoise=False)
    B = 100
    ctx = R.RenderContext(W, H, frames=B)
    stream = torch.cuda.current_stream()
    ref = None
    for rnd in range(3):
        for sset in settings:
            ctx.set_tuning("blocks_per_cu", 0)
            ctx.set_tuning("refill", 0)
            for kv in sset.split(","):
                k, v = kv.split("=")
                ctx.set_tuning(k, int(v))
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
            print("round %d %-28s raygen %.3f ms  traverse %.3f ms  shade %.3f ms per launch  same_bits=%s"
                  % (rnd, sset, kt["raygen_ms"], kt["traverse_ms"], kt["shade_ms"], same), flush=True)


if __name__ == "__main__":
    main()
