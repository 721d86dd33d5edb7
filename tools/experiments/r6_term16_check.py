import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import orc
import rt_octree_amd as R
from rt_octree_amd import synth
from helpers import cameras, oracle_frame
for depth, basis in ((7, 16), (9, 16), (10, 16)):
    tree = synth.make_tree(depth_limit=depth, basis_dim=basis, seed=5, shell=2.0)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    W, H, spp = 160, 120, 6
    pose = synth.orbit_poses(8)[2]
    ocam, cam = cameras(W, H, pose)
    ctx = R.RenderContext(W, H, frames=1)
    ctx.rng_seed()
    R.launch_renderer_batch(dt, [cam], R.RenderOptions(spp=spp, denoise=False), ctx, rng_jumps=[102])
    aux = ctx.download_aux()
    want = oracle_frame(ht, ocam, spp, frame=102)[0]
    d = aux.view(np.uint32) != want.view(np.uint32)
    print("depth %d: mismatching values per plane %s of %d pixels; alpha sum got %.1f want %.1f; wide nodes %d" % (
        depth, d.reshape(8, -1).sum(1).tolist(), W * H, aux[3].sum(), want[3].sum(), dt.wide_nodes))
