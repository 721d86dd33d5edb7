"""BASELINE.json configs[3] at size: 1920x1080, SPP 6, SH25 tree, GuidanceNet + filter, batched -- the shape of
the TanksAndTemple Truck run (T&T intrinsics fx = fy = 1160, a close orbit; a synthetic SH25 stand-in: no T&T data exists
offline).  Two COMPLETE 1080p frames against the oracle bit for bit (round 5: ~0.5 s each on the box's 16 cores),
size-independent properties of every pixel, oracle spot pixels on the third, batched == frame loop, the denoised image
against the oracle filter on the same maps."""
import ctypes as C

import numpy as np
import pytest

import orc
import rt_octree_amd as R
from helpers import assert_bits_equal, oracle_whole_frame
from rt_octree_amd import denoiser, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_c4_1080p_sh25_denoised_batch():
    W, H, spp = 1920, 1080, 6
    tree = synth.make_tree(depth_limit=8, basis_dim=25, seed=13, shell=2.0, radius=1.12)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    poses = synth.orbit_poses(8, radius=2.6)
    cams = []
    for p in poses[:3]:
        c = R.Camera(W, H, 1160.0, 1160.0)
        c.set_c2w(p)
        cams.append(c)
    opt = R.RenderOptions(spp=spp, denoise=True)
    ctx = R.RenderContext(W, H, frames=3)
    jumps = [100, 101, 102]
    R.launch_renderer_batch(dt, cams, opt, ctx, rng_jumps=jumps)
    # ---- every pixel: estimator properties
    for f in range(3):
        ctx.select_frame(f)
        aux = ctx.download_aux()
        noisy = ctx.download_image(noisy=True)
        a6 = aux[3] * spp
        assert np.all(np.abs(a6 - np.round(a6)) < 1e-5) and aux[3].min() >= 0 and aux[3].max() <= 1
        assert np.array_equal(aux[4:], aux[:4] * aux[:4])
        assert np.all(noisy[..., 3] == 1.0) and np.all(noisy[..., :3][aux[3] == 0] == 1.0)
        assert np.array_equal(noisy[..., :3], np.moveaxis(aux[:3], 0, -1))
        assert (aux[3] > 0).mean() > 0.2  # the close orbit fills a good part of the frame
        # ---- oracle spot pixels (bit-exact), spread over the frame incl. its corners
        rs = np.random.RandomState(f)
        idxs = list(rs.randint(0, W * H, 96)) + [0, W - 1, (H - 1) * W, H * W - 1, (H // 2) * W + W // 2]
        ocam = orc.camera(W, H, 1160.0, 1160.0, cams[f].transform.reshape(-1))
        oopt = orc.default_options(spp=spp)
        base = orc.rng(frame=jumps[f])
        for idx in idxs:
            a8, rgba = (C.c_float * 8)(), (C.c_float * 4)()
            assert orc.lib().orc_render_pixel(C.byref(ht.c), C.byref(ocam), C.byref(oopt), C.byref(base), int(idx), a8, rgba, None) == 0
            y, x = divmod(int(idx), W)
            assert_bits_equal(aux[:, y, x], np.array(a8[:], np.float32), "frame %d pixel %d" % (f, idx))
    # ---- two complete frames against the oracle: all 8 aux planes of all 2 073 600 pixels, and the RGBA8 bytes
    for f in (0, 2):
        aux_o, rgba_o = oracle_whole_frame(ht, cams[f], 1160.0, spp, jumps[f])
        ctx.select_frame(f)
        assert_bits_equal(ctx.download_aux(), aux_o, "whole 1080p frame %d: aux planes" % f)
        assert np.array_equal(ctx.download_rgba8(noisy=True), orc.rgba8(rgba_o)), "whole 1080p frame %d: RGBA8" % f
    # ---- batched == the reference's frame loop (one launch per frame, rng.advance in between)
    one = R.RenderContext(W, H)
    one.rng_seed()
    one.rng_advance(jumps[1] << 32)
    R.launch_renderer(dt, cams[1], opt, one)
    ctx.select_frame(1)
    assert_bits_equal(one.download_aux(), ctx.download_aux(), "frame loop vs batch")
    # ---- denoise: fused GuidanceNet -> exact filter == the oracle filter on the same maps; factorised within 1e-5
    torch.manual_seed(0)
    net = denoiser.FusedGuidanceNet(denoiser.GuidanceNetCompact.from_full(denoiser.GuidanceNet(8, 32, 5, 2, 4)).eval())
    aux_t = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
    wm, gm = net(aux_t[:3])
    ctx.select_frame(0)
    R.filtering(None, wm, gm, ctx.noisy_ptr, ctx.image_ptr)  # all three frames, one launch
    torch.cuda.synchronize()
    ctx.select_frame(2)
    exact = ctx.download_image()
    ref = orc.filter_levels(wm[2].cpu().numpy(), gm[2].cpu().numpy(), ctx.download_image(noisy=True))
    assert_bits_equal(exact, ref, "exact filter vs oracle at 1080p")
    ctx.select_frame(0)
    R.filtering(None, wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_FAST)
    torch.cuda.synchronize()
    ctx.select_frame(2)
    fast = ctx.download_image()
    assert np.allclose(fast[..., :3], exact[..., :3], rtol=2e-5, atol=2e-6)
    assert np.all((exact[..., :3] >= -1e-6) & (exact[..., :3] <= 1 + 1e-5))


def test_c4_bench_tree_one_complete_1080p_frame():
    """VERDICT r5 task 5: parity at BENCH size for configs[3] -- the very tree `bench.py --c4` times (depth 10, SH25, 4.03 M
    nodes, radius 1.12; its /dev/shm cache file is shared with the bench), the bench's cameras (T&T intrinsics, orbit radius
    2.6) and RNG jumps: a batched launch, ONE COMPLETE 1920x1080 frame against the oracle -- all 8 aux planes of all 2 073 600
    pixels and the RGBA8 bytes, bit for bit (volrend.cu:84-213 through orc.render_frame) -- and a second frame's spot pixels."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    args = bench.parse_args(["--c4"])
    path = bench.tree_cache_path(args, 0)
    if not os.path.exists(path):  # exactly bench.py's generation call
        th = synth.make_tree(depth_limit=args.depth, basis_dim=args.basis, shell=args.shell, radius=args.radius,
                             sdf=synth.scene_variant(0), seed=20230418)
        th.save_npz(path + ".tmp_test.npz")
        os.replace(path + ".tmp_test.npz", path)
        del th
    z = np.load(path)
    child, data, scale, offset, fmt = z["child"], z["data"], z["invradius3"], z["offset"], str(z["data_format"])
    assert fmt == "SH25" and child.shape[0] > 4_000_000
    ht = orc.HostTree(child, data, scale, offset, fmt)
    dt = R.N3Tree(path)  # the loader path the bench takes
    W, H, spp, fx = args.width, args.height, args.spp, args.fx
    assert (W, H, spp, fx) == (1920, 1080, 6, 1160.0)
    poses = synth.orbit_poses(bench.n_poses_for(1), radius=args.cam_radius)
    pick = [0, 57, 133]  # poses of the bench's launch groups
    cams = []
    for i in pick:
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(poses[i])
        cams.append(c)
    jumps = [bench.WARM_FRAMES_REF + i for i in pick]
    ctx = R.RenderContext(W, H, frames=len(pick))
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=spp, denoise=True), ctx, rng_jumps=jumps)
    aux_o, rgba_o = oracle_whole_frame(ht, cams[1], fx, spp, jumps[1])
    ctx.select_frame(1)
    aux = ctx.download_aux()
    assert (aux[3] > 0).mean() > 0.2
    assert_bits_equal(aux, aux_o, "bench C4 tree, pose %d: whole 1080p frame, aux planes" % pick[1])
    assert np.array_equal(ctx.download_rgba8(noisy=True), orc.rgba8(rgba_o)), "bench C4 tree: RGBA8 of the whole frame"
    ctx.select_frame(2)
    aux2 = ctx.download_aux()
    ocam = orc.camera(W, H, fx, fx, cams[2].transform.reshape(-1))
    oopt = orc.default_options(spp=spp)
    base = orc.rng(frame=jumps[2])
    rs = np.random.RandomState(7)
    for idx in list(rs.randint(0, W * H, 128)) + [0, W * H - 1, (H // 2) * W + W // 2]:
        a8, rgba = (C.c_float * 8)(), (C.c_float * 4)()
        assert orc.lib().orc_render_pixel(C.byref(ht.c), C.byref(ocam), C.byref(oopt), C.byref(base), int(idx), a8, rgba, None) == 0
        y, x = divmod(int(idx), W)
        assert_bits_equal(aux2[:, y, x], np.array(a8[:], np.float32), "bench C4 tree pose %d pixel %d" % (pick[2], idx))
    ctx.free()
    dt.free()
