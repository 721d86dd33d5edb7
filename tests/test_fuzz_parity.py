"""Seeded random scenes through every traversal kernel against the oracle, bit for bit: random tree
depth / SH order / format, ragged image sizes, cameras outside, inside and grazing the box, random
crop boxes, basis masks, backgrounds, step sizes, sigma thresholds, SPP, batch sizes -- the corners the
fixed-seed parity tests do not enumerate."""
import numpy as np
import pytest

import orc
import rt_octree_amd as R
from rt_octree_amd import synth

from helpers import assert_bits_equal, rgba_tree

pytestmark = pytest.mark.gpu


def _scene(seed):
    rs = np.random.RandomState(1000 + seed)
    basis = int(rs.choice([4, 9, 16, 25]))
    tree = synth.make_tree(depth_limit=int(rs.randint(3, 8)), basis_dim=basis, seed=int(rs.randint(1, 1 << 20)),
                           shell=float(rs.uniform(0.8, 2.0)))
    if rs.rand() < 0.2:
        tree = rgba_tree(tree, seed=seed)
    W, H = int(rs.randint(9, 150)), int(rs.randint(9, 110))
    fx = float(rs.uniform(0.5, 2.5) * W)
    n = int(rs.randint(1, 6))
    poses = []
    for _ in range(n):
        mode = rs.rand()
        if mode < 0.2:  # inside the scene box
            pos = rs.uniform(-0.6, 0.6, 3)
        elif mode < 0.35:  # far away, small on screen
            pos = rs.randn(3)
            pos = pos / np.linalg.norm(pos) * rs.uniform(8, 20)
        else:
            pos = rs.randn(3)
            pos = pos / np.linalg.norm(pos) * rs.uniform(2.0, 5.0)
        target = rs.uniform(-0.5, 0.5, 3) if rs.rand() < 0.8 else rs.uniform(-3, 3, 3)  # sometimes looks past it
        poses.append(synth.look_at_c2w(pos, target))
    opt = {"spp": int(rs.choice(R.SUPPORTED_SPP)), "background_brightness": float(rs.choice([1.0, 0.0, rs.rand()])),
           "step_size": float(rs.choice([1e-4, 1e-3, 0.02])), "sigma_thresh": float(rs.choice([1e-2, 0.0, 0.5]))}
    if rs.rand() < 0.4:
        lo = rs.uniform(0.0, 0.4, 3)
        hi = rs.uniform(0.6, 1.0, 3)
        opt["render_bbox"] = [float(v) for v in np.concatenate([lo, hi])]
    if rs.rand() < 0.3 and tree.data_format != "RGBA":
        a = int(rs.randint(0, basis))
        opt["basis_minmax"] = [a, int(rs.randint(a, basis))]
    return tree, W, H, fx, poses, opt, int(rs.randint(0, 300))


def _seeds():
    """32 scenes by default; RTO_FUZZ_SEEDS="first:last" widens the sweep for a soak run (tools/fuzz_soak.sh)"""
    import os
    spec = os.environ.get("RTO_FUZZ_SEEDS", "")
    if ":" in spec:
        a, b = spec.split(":")
        return range(int(a), int(b))
    return range(32)


@pytest.mark.parametrize("seed", _seeds())
def test_random_scene_all_kernels_bit_exact(seed):
    tree, W, H, fx, poses, optkw, frame0 = _scene(seed)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    spp = optkw.pop("spp")
    cams, want = [], []
    for i, p in enumerate(poses):
        cam = R.Camera(W, H, fx, fx)
        cam.set_c2w(p)
        cams.append(cam)
        ocam = orc.camera(W, H, fx, fx, cam.transform.reshape(-1))
        aux, rgba, _ = orc.render_frame(ht, ocam, orc.default_options(spp=spp, **optkw), orc.rng(frame=frame0 + i))
        want.append((aux, rgba))
    opt = R.RenderOptions(spp=spp, denoise=False, **optkw)
    ctx = R.RenderContext(W, H, frames=len(cams))
    R.launch_renderer_batch(dt, cams, opt, ctx, rng_jumps=[frame0 + i for i in range(len(cams))])
    for i in range(len(cams)):
        ctx.select_frame(i)
        assert_bits_equal(ctx.download_aux(), want[i][0], "batched aux %d" % i)
        assert_bits_equal(ctx.download_image(), want[i][1], "batched image %d" % i)
    one = R.RenderContext(W, H)
    for kernel in (R.KERNEL_FAST, R.KERNEL_GENERIC):
        one.set_kernel(kernel)
        one.rng_seed()
        one.rng_advance(frame0 << 32)
        R.launch_renderer(dt, cams[0], opt, one)
        assert_bits_equal(one.download_aux(), want[0][0], "kernel %d aux" % kernel)
