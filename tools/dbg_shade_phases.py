"""Where a shading wave's lifetime goes (debug build: bash tools/dbg_build.sh, RTO_LIB=rt-octree_amd/lib_dbg/librto.so): mean shader
clocks from a wave's start to the end of each phase of shade_kernel, for the waves that hold hit entries and for the others.
python3 tools/dbg_shade_phases.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import synth  # noqa: E402

args = bench.parse_args([])
path = bench.tree_cache_path(args)
if not os.path.exists(path):
    synth.make_tree(depth_limit=10, basis_dim=16, shell=2.5).save_npz(path)
dt = R.N3Tree(path)
W = H = 800
fx = synth.blender_focal(W)
B = 100
cams = []
for p in synth.orbit_poses(200)[:B]:
    c = R.Camera(W, H, fx, fx)
    c.set_c2w(p)
    cams.append(c)
ctx = R.RenderContext(W, H, frames=B)
ctx.set_lean_outputs(True)
import numpy as np  # noqa: E402
L = C.CDLL(R.LIB_PATH)
buf = np.zeros((1 << 19, 8), np.uint64)
for rep in range(3):
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=True), ctx, rng_jumps=[100 + i for i in range(B)])
    torch.cuda.synchronize()
    assert L.rto_debug_shade_phases(buf.ctypes.data_as(C.c_void_p), 1) == 0
w = buf[buf[:, 7] == 1].astype(np.int64)
t = w[:, 1:6] - w[:, 0:1]
hit = w[:, 6] > 0
names = ["tile marks + hit lists loaded, prefix sum", "entries published (last window)", "entries shaded (last window)",
         "pixel sums done", "stores retired (wave ends)"]
span = (w[:, 5].max() - w[:, 0].min())
print("shade_kernel<6,2,49>, 100 C2 frames: %d waves with hit entries (%.1f entries each; %d of them with more than one window), %d without; "
      "kernel span %d ticks" % (hit.sum(), w[hit, 6].mean(), (w[:, 6] > 320).sum(), (~hit).sum(), span))
for sel, nm in ((hit & (w[:, 6] <= 320), "hit waves, one window"), (w[:, 6] > 320, "hit waves, two or more windows"), (~hit, "waves without entries")):
    if sel.sum() == 0:
        continue
    m = t[sel].mean(axis=0)
    print("%s (%d waves, %.0f entries): %s" % (nm, sel.sum(), w[sel, 6].mean(),
          "  ".join("%s at %.0f" % (n, v) for n, v in zip(names, m))))
    print("   wave-ticks total %.3e (= %.1f slots busy over the span)" % (t[sel, 4].sum(), t[sel, 4].sum() / max(span, 1)))
