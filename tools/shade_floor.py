"""What bounds shade_kernel on the bench workload?  Its time per 100-frame launch for (a) the bench orbit, (b) the same
cameras turned away from the model (every tile culled: the kernel only writes 48 B of background per pixel -- its store
floor), (c) culling off on the orbit (every pixel reads its list).  python tools/shade_floor.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import synth  # noqa: E402


def main():
    import torch
    cache = "/dev/shm/rto_ab_tree_d10_b16.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        dt = R.N3Tree.from_arrays(z["child"], z["data"], z["scale"], z["offset"], str(z["data_format"]))
    else:
        tree = synth.make_tree(depth_limit=10, basis_dim=16, shell=2.5)
        np.savez(cache + ".tmp.npz", child=tree.child, data=tree.data, scale=tree.scale, offset=tree.offset, data_format=tree.data_format)
        os.replace(cache + ".tmp.npz", cache)
        dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    W = H = 800
    fx = synth.blender_focal(W)
    B = 100
    poses = synth.orbit_poses(200)[:B]

    def cams_of(ps):
        out = []
        for p in ps:
            c = R.Camera(W, H, fx, fx)
            c.set_c2w(p)
            out.append(c)
        return out

    away = poses.copy()
    away[:, :3, 0] *= -1.0  # turn the camera round its up axis: x and z axes flipped
    away[:, :3, 2] *= -1.0
    opt = R.RenderOptions(spp=6, denoise=True)
    ctx = R.RenderContext(W, H, frames=B)
    stream = torch.cuda.current_stream()
    for rnd in range(2):
        for name, cams, cull in (("orbit", cams_of(poses), 1), ("away (all culled)", cams_of(away), 1), ("orbit, culling off", cams_of(poses), 0)):
            ctx.set_tuning("cull", cull)
            ctx.kernel_timing(True)
            for rep in range(3):
                ctx.rng_seed()
                R.launch_renderer_batch(dt, cams, opt, ctx, stream, rng_jumps=[100 + k for k in range(B)])
            torch.cuda.synchronize()
            kt = ctx.kernel_timing_read()
            live, total = ctx.queue_stats()
            print("round %d %-22s tiles marched %.3f  thresholds %.3f  traverse %.3f  shade %.3f ms per launch"
                  % (rnd, name, live / total, kt["raygen_ms"], kt["traverse_ms"], kt["shade_ms"]), flush=True)


if __name__ == "__main__":
    main()
