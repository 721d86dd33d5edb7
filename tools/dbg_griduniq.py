"""Debug build (make EXTRA=-DRTO_DBG_COUNTERS\\ -DRTO_DBG_GRIDUNIQ): distinct top-grid cells and distinct 64-byte
nodew lines per wave-level load of the persistent kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
R.lib().rto_debug_zero_queue(ctx._h)
R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=False), ctx, rng_jumps=list(range(100, 108)))
torch.cuda.synchronize()
out = (C.c_uint64 * 8)()
R.lib().rto_debug_read_queue(ctx._h, out)
print("grid : %d wave-loads, %.2f distinct cells, %.1f lanes each" % (out[2], out[3] / max(out[2], 1), out[4] / max(out[2], 1)))
print("nodew: %d wave-loads, %.2f distinct 64-B lines, %.1f lanes each" % (out[5], out[6] / max(out[5], 1), out[7] / max(out[5], 1)))
