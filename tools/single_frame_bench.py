"""The per-frame operator alone (rto_launch_renderer, one launch + one host wait per frame: the reference's loop shape,
main_headless.cpp:485-506) under the single-frame tuning keys: tools/single_frame_bench.py strip_rows=1 strip_rows=2 ...
-> mean ms per frame over 96 poses of the bench scene, images compared bit for bit with the first setting."""
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
    settings = [a for a in sys.argv[1:] if "=" in a] or ["strip_rows=1", "strip_rows=2", "strip_rows=4"]
    cache = "/dev/shm/rto_ab_tree_d10_b16.npz"
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
    for p in synth.orbit_poses(200)[:96]:
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(p)
        cams.append(c)
    opt = R.RenderOptions(spp=6, denoise=False)
    ctx = R.RenderContext(W, H)
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ref = None
    for rnd in range(2):
        for sset in settings:
            for kv in sset.split(","):
                k, v = kv.split("=")
                ctx.set_tuning(k, int(v))
            tot = 0.0
            t0 = time.perf_counter()
            for i, cam in enumerate(cams):
                ctx.rng_seed()
                ctx.rng_advance((100 + i) << 32)
                e0.record(stream)
                R.launch_renderer(dt, cam, opt, ctx, stream)
                e1.record(stream)
                e1.synchronize()
                tot += e0.elapsed_time(e1)
            wall = time.perf_counter() - t0
            aux = ctx.download_aux()
            if ref is None:
                ref = aux
            print("round %d %-24s render %.4f ms per frame (events), %.0f frames/s wall  same_bits=%s" % (
                rnd, sset, tot / len(cams), len(cams) / wall, np.array_equal(aux.view(np.uint32), ref.view(np.uint32))), flush=True)


if __name__ == "__main__":
    main()
