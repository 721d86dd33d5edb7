"""Per-branch occupancy of render_persist's march loop (debug build: bash tools/dbg_build.sh, RTO_LIB=rt-octree_amd/lib_dbg/librto.so).
For every branch of the loop body: how often a wave executes it and with how many lanes -- the numbers behind
`lanes_per_valu_inst` (VERDICT r3 task 3).  python3 tools/dbg_counters.py [tuning k=v ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
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
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_tuning(k, int(v))
ctx.rng_seed()
R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=False), ctx, rng_jumps=[100 + i for i in range(B)])
torch.cuda.synchronize()
out = (C.c_uint64 * 24)()
R.lib().rto_debug_read_queue(ctx._h, out)
names = ["iteration (any active lane)", "descend one level", "leaf: march step", "sigma > thresh", "hit: threshold crossed",
         "restart (ray goes on)", "ray set-up (refill)", "top-grid lookup"]
wv = [out[17]] + [out[i] for i in range(1, 8)]
ln = [out[18]] + [out[8 + i] for i in range(1, 8)]
print("per 100-frame launch, C2 scene%s" % (" [" + " ".join(sys.argv[1:]) + "]" if sys.argv[1:] else ""))
for i, nm in enumerate(names):
    print("%-30s wave executions %8.2f M (%.3f of iterations)   lanes %9.1f M   lanes per execution %5.1f" % (
        nm, wv[i] / 1e6, wv[i] / max(wv[0], 1), ln[i] / 1e6, ln[i] / max(wv[i], 1)))
print("loads per march step %.3f   march steps %.1f M" % ((ln[1] + ln[2]) / max(ln[2], 1), ln[2] / 1e6))
