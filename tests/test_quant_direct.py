"""Quantised PlenOctrees rendered straight from their codebooks (RTO_TREE_QUANT_DIRECT, SURVEY 8f
rank 2) must give the very pixels of the reference's route -- N3Tree::load_npz expanding the set to
dense fp16 (n3tree.cpp:279-340) and the kernel reading that -- because both feed the same fp16
coefficients into the same sums.  Bit-exact: against the decoded tree through the HIP path and
against the oracle rendering the decoded arrays."""
import os

import numpy as np
import pytest

import orc
import rt_octree_amd as R
from rt_octree_amd import synth

from helpers import assert_bits_equal, cameras

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cams(W, H, n):
    out = []
    for i in range(n):
        _, cam = cameras(W, H, synth.orbit_poses(8)[i])
        out.append(cam)
    return out


@pytest.mark.parametrize("basis,n_retain,spp,quantiser", [(9, 1, 6, "luminance"), (4, 0, 4, "luminance"), (16, 2, 6, "luminance"),
                                                          (25, 1, 2, "luminance"), (16, 1, 6, "median_cut")])
def test_direct_equals_decoded_and_oracle(tmp_path, basis, n_retain, spp, quantiser):
    """(round 6: also for codebooks made by a real median cut over the leaves of positive density, compress_octree.py:68-119)"""
    W, H = 96, 64
    tree = synth.make_tree(depth_limit=6, basis_dim=basis, seed=11)
    path = str(tmp_path / "tree.npz")
    decoded = tree.save_quant_npz(path, n_retain=n_retain, quantiser=quantiser)
    direct = R.N3Tree(path, quant_direct=True)
    dense = R.N3Tree(path)
    assert direct.data_dim == dense.data_dim == 3 * basis + 1  # (the footprint only wins once slots outnumber codebook entries)
    ht = orc.HostTree(tree.child, decoded, tree.scale, tree.offset, tree.data_format)
    cams = _cams(W, H, 3)
    opt = R.RenderOptions(spp=spp, denoise=False)
    a, b = R.RenderContext(W, H, frames=3), R.RenderContext(W, H, frames=3)
    R.launch_renderer_batch(direct, cams, opt, a, rng_jumps=[100, 101, 102])
    R.launch_renderer_batch(dense, cams, opt, b, rng_jumps=[100, 101, 102])
    for f in range(3):
        a.select_frame(f)
        b.select_frame(f)
        assert_bits_equal(a.download_aux(), b.download_aux(), "aux f%d" % f)
        assert_bits_equal(a.download_image(), b.download_image(), "image f%d" % f)
    ocam = orc.camera(W, H, cams[1].fx, cams[1].fy, cams[1].transform.reshape(-1))
    aux_o, rgba_o, _ = orc.render_frame(ht, ocam, orc.default_options(spp=spp), orc.rng(frame=101))
    a.select_frame(1)
    assert_bits_equal(a.download_aux(), aux_o, "aux vs oracle")
    assert_bits_equal(a.download_image(), rgba_o, "image vs oracle")


def test_single_frame_entry_point_routes_quant_trees(tmp_path):
    """launch_renderer on a direct tree = a batch of one into the selected slot, same RNG stream."""
    W, H, spp = 80, 56, 6
    tree = synth.make_tree(depth_limit=5, basis_dim=9, seed=5)
    path = str(tmp_path / "tree.npz")
    tree.save_quant_npz(path, n_retain=1)
    direct, dense = R.N3Tree(path, quant_direct=True), R.N3Tree(path)
    cam = _cams(W, H, 2)[1]
    opt = R.RenderOptions(spp=spp, denoise=True)
    a, b = R.RenderContext(W, H, frames=2), R.RenderContext(W, H)
    a.select_frame(1)
    for ctx, t in ((a, direct), (b, dense)):
        ctx.rng_seed()
        ctx.rng_advance()
        R.launch_renderer(t, cam, opt, ctx)
    assert_bits_equal(a.download_aux(), b.download_aux(), "aux")
    assert_bits_equal(a.download_image(noisy=True), b.download_image(noisy=True), "noisy")
    a.set_kernel(R.KERNEL_GENERIC)
    with pytest.raises(R.RtoError):
        R.launch_renderer(direct, cam, opt, a)


def test_golden_quant_file_direct():
    """The committed quantised fixture (n_retain = 7 of 9, random maps) through both routes."""
    path = os.path.join(GOLD, "npz_quant.npz")
    direct, dense = R.N3Tree(path, quant_direct=True), R.N3Tree(path)
    W, H = 64, 64
    cams = _cams(W, H, 2)
    opt = R.RenderOptions(spp=6, denoise=False)
    a, b = R.RenderContext(W, H, frames=2), R.RenderContext(W, H, frames=2)
    R.launch_renderer_batch(direct, cams, opt, a)
    R.launch_renderer_batch(dense, cams, opt, b)
    for f in range(2):
        a.select_frame(f)
        b.select_frame(f)
        assert_bits_equal(a.download_image(), b.download_image(), "image f%d" % f)
    assert float(np.abs(a.download_image()[..., :3]).max()) > 0


def test_shuffled_quant_tree_is_relaid_like_any_other(tmp_path):
    """a quantised file whose nodes are stored in a random order: the breadth-first relayout at upload gathers
    quant_map / sigma / data_retained per slot too -- direct and expanded routes still equal the ordered tree."""
    W, H = 64, 48
    tree = synth.make_tree(depth_limit=5, basis_dim=9, seed=8)
    shuffled = synth.shuffle_nodes(tree, seed=2)
    p0, p1 = str(tmp_path / "ordered.npz"), str(tmp_path / "shuffled.npz")
    tree.save_quant_npz(p0, n_retain=2)
    # the stand-in quantiser ranks slots by luminance: the same slots get the same codes in either storage order
    shuffled.save_quant_npz(p1, n_retain=2)
    cams = _cams(W, H, 2)
    opt = R.RenderOptions(spp=6, denoise=False)
    ref = R.RenderContext(W, H, frames=2)
    R.launch_renderer_batch(R.N3Tree(p0), cams, opt, ref)
    for direct in (False, True):
        ctx = R.RenderContext(W, H, frames=2)
        R.launch_renderer_batch(R.N3Tree(p1, quant_direct=direct), cams, opt, ctx)
        for f in range(2):
            ref.select_frame(f)
            ctx.select_frame(f)
            assert_bits_equal(ctx.download_aux(), ref.download_aux(), "direct %s frame %d" % (direct, f))
