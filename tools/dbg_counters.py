"""Reads the RTO_DBG_COUNTERS words of a debug build (make EXTRA=-DRTO_DBG_COUNTERS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rt_octree_amd as R
from rt_octree_amd import synth
tree = synth.make_tree(depth_limit=10, basis_dim=16, shell=2.5)
dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
W = H = 800
fx = synth.blender_focal(W)
cams = []
for p in synth.orbit_poses(200)[:8]:
    c = R.Camera(W, H, fx, fx); c.set_c2w(p); cams.append(c)
ctx = R.RenderContext(W, H, frames=8)
for rf in (0,):
    ctx.set_tuning("refill", rf)
    R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=False), ctx, rng_jumps=list(range(100, 108)))
    torch.cuda.synchronize()
    # queue words live in the ctx; fetch them through a raw hipMemcpy via torch
    hip = C.CDLL("libamdhip64.so")
    # the queue pointer is not exported; counters are read by the debug ABI below
    out = (C.c_uint64 * 8)()
    R.lib().rto_debug_read_queue(ctx._h, out)
    ws, ls, ll, lf = out[2], out[3], out[4], out[5]
    print("refill %3d: wave_iters/frame %.0f  lane_iters/frame %.0f  util %.3f  loads/frame %.0f leafs/frame %.0f  cousin steps/frame %.0f  sibling steps/frame %.0f" % (
        rf, ws / 8, ls / 8, ls / (64.0 * ws), ll / 8, lf / 8, out[6] / 8, out[7] / 8))
    R.lib().rto_debug_zero_queue(ctx._h)
