"""A tree with more than 2^27 leaf slots (a full octree of depth 9: 19.2 M nodes, 153 M slots): round 1 refused the
fast / batched kernels above 2^27 slots; hit-list entries now carry 31 - ceil(log2 SPP) slot bits (28 at SPP <= 8) and
top-grid entries 29, and a launch whose SPP leaves too few bits (SPP 32 here) falls back to the generic kernel."""
import numpy as np
import pytest

import orc
import rt_octree_amd as R
from helpers import assert_bits_equal, cameras
from rt_octree_amd import synth

pytestmark = pytest.mark.gpu


def _full_octree(depth, seed=3):
    """every cell refined down to `depth` node levels; breadth-first: the children of node i are 8 i + 1 .. 8 i + 8"""
    n_internal = (8 ** (depth - 1) - 1) // 7
    cap = (8 ** depth - 1) // 7
    child = np.zeros((cap, 8), np.int32)
    i = np.arange(n_internal, dtype=np.int64)[:, None]
    child[:n_internal] = (7 * i + 1 + np.arange(8)[None, :]).astype(np.int32)  # relative offsets
    rs = np.random.RandomState(seed)
    data = np.zeros((cap, 8, 4), np.float16)
    n_leaf_nodes = cap - n_internal
    occ = rs.rand(n_leaf_nodes, 8) < 0.004
    data[n_internal:, :, 3] = (occ * rs.uniform(20, 400, (n_leaf_nodes, 8))).astype(np.float16)
    data[n_internal:, :, :3] = rs.rand(n_leaf_nodes, 8, 3).astype(np.float16)
    return synth.SynthTree(child.reshape(cap, 2, 2, 2), data.reshape(cap, 2, 2, 2, 4), np.full(3, 1 / 3.0, np.float32),
                           np.full(3, 0.5, np.float32), "RGBA", depth, {})


def test_tree_beyond_2_pow_27_slots():
    tree = _full_octree(9)
    assert tree.capacity * 8 > (1 << 27)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    W, H = 40, 32
    ocam, cam = cameras(W, H, synth.orbit_poses(5)[2])
    for spp in (6, 32):  # 6: fast and batched kernels (28 slot bits); 32: 26 bits are too few -> generic kernel
        want, rgba_o, st = orc.render_frame(ht, ocam, orc.default_options(spp=spp), orc.rng(frame=3))
        assert st["hit_rays"] > 100
        ctx = R.RenderContext(W, H, frames=2)
        ctx.rng_seed()
        R.launch_renderer_batch(dt, [cam, cam], R.RenderOptions(spp=spp, denoise=False), ctx, rng_jumps=[0, 3])
        ctx.select_frame(1)
        assert_bits_equal(ctx.download_aux(), want, "batched, spp %d" % spp)
        one = R.RenderContext(W, H)
        one.rng_seed()
        one.rng_advance(3 << 32)
        R.launch_renderer(dt, cam, R.RenderOptions(spp=spp, denoise=False), one)
        assert_bits_equal(one.download_aux(), want, "single frame, spp %d" % spp)
        if spp == 32:
            one.set_kernel(R.KERNEL_FAST)
            with pytest.raises(R.RtoError):
                R.launch_renderer(dt, cam, R.RenderOptions(spp=spp, denoise=False), one)
