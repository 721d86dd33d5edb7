"""Parity on the EXACT workload bench.py times (BASELINE.json configs[1] and configs[4]; VERDICT r2 task 2):
the bench tree (synth.make_tree(depth_limit=10, basis_dim=16, shell=2.5): SH16, 2.12 M nodes), 800x800, the bench's
orbit poses and RNG jumps, a 100-frame rto_launch_renderer_batch -- SPP 6 and SPP 1.

Per batch (round 5, VERDICT r4 task 2: the oracle renders an 800x800 frame of this tree in ~0.13 s on the 16 cores a GPU box
grants -- rounds 1-3 measured 0.8 frames/s because 256 throttled threads shared them -- so whole frames are affordable):
  * TWO COMPLETE FRAMES per configuration against orc.render_frame, all 8 aux planes and the RGBA8 bytes, bit for bit
    (the oracle's render_kernel + trace_ray, volrend.cu:84-213, rt_core.cuh:195-332);
  * >= 256 oracle spot pixels per further checked frame, bit for bit (orc_render_pixel);
  * the estimator's size-independent properties on EVERY pixel of EVERY frame (on the device, over the zero-copy
    views of the batch buffers);
  * batch slot == the reference's frame loop (one launch per frame, rng.advance in between);
  * exact filter == the oracle filter on one frame; the factorised filter within its stated tolerance, and the
    number of RGBA8 bytes by which the two routes differ (main_headless.cpp:535-538 truncation)."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import rt_octree_amd as R
from helpers import assert_bits_equal, oracle_whole_frame
from rt_octree_amd import denoiser, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

W = H = 800
B = 100
WARM = 100  # main_headless.cpp:469-479: frame i renders with the RNG advanced 100 + i times
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench_scene():
    """the bench tree on host (oracle) and device, and the bench's cameras"""
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    cache = os.path.join(base, "rto_test_bench_tree_d10_b16.npz")
    if os.path.exists(cache):
        z = np.load(cache)
        child, data, scale, offset, fmt = z["child"], z["data"], z["scale"], z["offset"], str(z["data_format"])
    else:
        t = synth.make_tree(depth_limit=10, basis_dim=16, shell=2.5)
        child, data, scale, offset, fmt = t.child, t.data, t.scale, t.offset, t.data_format
        np.savez(cache + ".tmp.npz", child=child, data=data, scale=scale, offset=offset, data_format=fmt)
        os.replace(cache + ".tmp.npz", cache)
    assert fmt == "SH16" and child.shape[0] > 2_000_000
    ht = orc.HostTree(child, data, scale, offset, fmt)
    dt = R.N3Tree.from_arrays(child, data, scale, offset, fmt)
    fx = synth.blender_focal(W)
    poses = synth.orbit_poses(200)
    cams = []
    for p in poses:
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(p)
        cams.append(c)
    yield ht, dt, cams, fx
    dt.free()


def spot_pixels(ht, cam, fx, spp, jump, aux, n, seed):
    """n random pixels + the corners and the centre of one frame against the oracle; -> pixels checked"""
    ocam = orc.camera(W, H, fx, fx, cam.transform.reshape(-1))
    oopt = orc.default_options(spp=spp)
    base = orc.rng(frame=jump)
    rs = np.random.RandomState(seed)
    idxs = list(rs.randint(0, W * H, n)) + [0, W - 1, (H - 1) * W, H * W - 1, (H // 2) * W + W // 2]
    hit = 0
    for idx in idxs:
        a8, rgba = (C.c_float * 8)(), (C.c_float * 4)()
        assert orc.lib().orc_render_pixel(C.byref(ht.c), C.byref(ocam), C.byref(oopt), C.byref(base), int(idx), a8, rgba, None) == 0
        y, x = divmod(int(idx), W)
        assert_bits_equal(aux[:, y, x], np.array(a8[:], np.float32), "jump %d pixel %d" % (jump, idx))
        hit += a8[3] > 0
    assert hit > n // 10, "the spot pixels must not all be background (%d of %d hit)" % (hit, len(idxs))
    return len(idxs)


def device_properties(ctx, spp, n, denoise=False):
    """estimator properties of every pixel of the first n frame slots, evaluated on the device"""
    aux_v, noisy_v, image_v = ctx.batch_views()
    aux = torch.as_tensor(aux_v, device="cuda:0")[:n]
    # the rendered RGBA: the noisy buffer when a denoise stage follows, else the final image (volrend.cu:206)
    noisy = torch.as_tensor(noisy_v if denoise else image_v, device="cuda:0")[:n]
    alpha = aux[:, 3]
    a = alpha * spp
    assert bool(torch.all((a - torch.round(a)).abs() < 1e-5)) and float(alpha.min()) >= 0 and float(alpha.max()) <= 1
    assert bool(torch.equal(aux[:, 4:], aux[:, :4] * aux[:, :4]))                  # squares planes (volrend.cu:195-202)
    assert bool(torch.all(noisy[..., 3] == 1.0))                                   # alpha forced to 1 (:205)
    assert bool(torch.equal(noisy[..., :3], aux[:, :3].permute(0, 2, 3, 1)))       # image == aux colour planes
    rgb = aux[:, :3]
    assert bool(torch.all((rgb >= 0) & (rgb <= 1 + 1e-6)))
    miss = (alpha == 0).unsqueeze(1).expand(-1, 3, -1, -1)
    assert bool(torch.all(rgb[miss] == 1.0))                                       # misses = background (:174-178)
    cover = (alpha > 0).float().mean(dim=(1, 2))
    assert float(cover.min()) > 0.1 and float(cover.max()) < 0.6                   # every pose sees the model
    return float(cover.mean())


@pytest.mark.parametrize("spp", [6, 1])
def test_bench_batch_of_100_frames_matches_the_oracle(bench_scene, spp):
    ht, dt, cams, fx = bench_scene
    opt = R.RenderOptions(spp=spp, denoise=False)
    ctx = R.RenderContext(W, H, frames=B)
    ctx.rng_seed()
    jumps = [WARM + i for i in range(B)]
    R.launch_renderer_batch(dt, cams[:B], opt, ctx, rng_jumps=jumps)
    device_properties(ctx, spp, B)
    checked = 0
    for f in (0, 37, 99) if spp == 6 else (0, 63):
        ctx.select_frame(f)
        checked += spot_pixels(ht, cams[f], fx, spp, jumps[f], ctx.download_aux(), 256, seed=1000 * spp + f)
    assert checked >= 2 * 261
    # two complete frames against the oracle: every aux plane of every pixel, and the RGBA8 bytes (main_headless.cpp:535-538)
    for f in (1, 98):
        aux_o, rgba_o = oracle_whole_frame(ht, cams[f], fx, spp, jumps[f])
        ctx.select_frame(f)
        assert_bits_equal(ctx.download_aux(), aux_o, "whole frame %d, spp %d: aux planes" % (f, spp))
        assert np.array_equal(ctx.download_rgba8(), orc.rgba8(rgba_o)), "whole frame %d, spp %d: RGBA8" % (f, spp)
    # batch slot == the frame loop (main_headless.cpp:485-506): launch_renderer; ctx.rng.advance()
    one = R.RenderContext(W, H)
    for f in (5, 99):
        one.rng_seed()
        one.rng_advance(jumps[f] << 32)
        R.launch_renderer(dt, cams[f], opt, one)
        ctx.select_frame(f)
        assert_bits_equal(one.download_aux(), ctx.download_aux(), "frame loop vs batch slot %d" % f)
        assert np.array_equal(one.download_rgba8(), ctx.download_rgba8())
    # the second half of the reference's 200-pose loop through the same context (queues re-armed)
    R.launch_renderer_batch(dt, cams[B:2 * B], opt, ctx, rng_jumps=[WARM + B + i for i in range(B)])
    device_properties(ctx, spp, B)
    ctx.select_frame(42)
    spot_pixels(ht, cams[B + 42], fx, spp, WARM + B + 42, ctx.download_aux(), 256, seed=7)
    one.free()
    ctx.free()


def test_bench_denoise_routes_on_the_bench_frame(bench_scene):
    """the two filter routes bench.py can time, on frames of the bench workload with the bench's trained network:
    exact == oracle bit for bit; factorised (the headline route) within 2e-5 relative; RGBA8 bytes that differ counted"""
    ht, dt, cams, fx = bench_scene
    n = 4
    opt = R.RenderOptions(spp=6, denoise=True)
    ctx = R.RenderContext(W, H, frames=n)
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams[:n], opt, ctx, rng_jumps=[WARM + i for i in range(n)])
    torch.manual_seed(0)
    full = denoiser.GuidanceNet(8, 32, 5, 2, 4)
    wpath = os.path.join(ROOT, "rt-octree_amd", "weights", "guidance_synth_lego.pt")
    if os.path.exists(wpath):
        full.load_state_dict(torch.load(wpath, map_location="cpu"))
    net = denoiser.FusedGuidanceNet(denoiser.GuidanceNetCompact.from_full(full).eval())
    aux_t = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
    wm, gm = net(aux_t[:n], squares_implied=True)
    ctx.select_frame(0)
    R.filtering(None, wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_EXACT)
    torch.cuda.synchronize()
    ctx.select_frame(2)
    exact = ctx.download_image()
    exact8 = ctx.download_rgba8()
    ref = orc.filter_levels(wm[2].cpu().numpy(), gm[2].cpu().numpy(), ctx.download_image(noisy=True))
    assert_bits_equal(exact, ref, "exact filter vs oracle on the bench frame")
    assert np.array_equal(exact8, orc.rgba8(ref))
    ctx.select_frame(0)
    R.filtering(None, wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_FAST)
    torch.cuda.synchronize()
    ctx.select_frame(2)
    fast = ctx.download_image()
    fast8 = ctx.download_rgba8()
    assert np.allclose(fast[..., :3], exact[..., :3], rtol=2e-5, atol=2e-6)
    # (uint8_t)(f * 255) truncates: a 2e-5 relative difference may step over an integer boundary.  Counted, bounded,
    # and never by more than one code value (README / DESIGN say the headline route is the tolerance route).
    diff = fast8.astype(np.int16) - exact8.astype(np.int16)
    assert np.abs(diff).max() <= 1
    assert np.count_nonzero(diff) <= 2000, np.count_nonzero(diff)  # of 2.56 M bytes
    # the packed route (fp16 maps between the two kernels) is the factorised route on the same values
    net.forward_packed(aux_t[:n], squares_implied=True)
    ctx.select_frame(0)
    net.filter_packed(ctx.noisy_ptr, ctx.image_ptr)
    torch.cuda.synchronize()
    ctx.select_frame(2)
    assert_bits_equal(ctx.download_image(), fast, "packed maps vs fp32 planes through the factorised filter")
    ctx.free()


def test_c3_eight_scenes_sharded_over_eight_ranks_both_scene_maps():
    """BASELINE.json configs[2]: 8 scenes x 200 poses, 800x800 SPP 6, frames sharded over 8 ranks -- on ONE GPU, rank after
    rank: each of the 8 ranks' first launch groups under both scene maps of bench.py (pose: frame g -> rank g mod 8, every rank
    renders every scene; scene: scene s -> rank s mod 8), and for every scene one COMPLETE frame of some rank's group against
    the oracle.  A frame is (scene, pose) with the RNG advanced 100 + pose times: it must not depend on the rank, the map or
    the slot it lands in.  (Scenes: bench.py's 8 variants; two of them at the bench's own depth 10 -- 2.1 M nodes -- the other six at
    depth 9, ~0.5 M nodes each, so that 8 trees are generated and uploaded in test time.)"""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    world, n_scenes, n_poses, Bc = 8, 8, 200, 8
    fx = synth.blender_focal(W)
    poses = synth.orbit_poses(n_poses)
    cams = []
    for p in poses:
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(p)
        cams.append(c)
    hts, dts = [], []
    for s in range(n_scenes):
        # (VERDICT r5 task 5: scenes 0 and 5 at the bench's depth 10 -- 2.1 M nodes, what `bench.py --scenes 8` renders -- the rest at 9)
        t = synth.make_tree(depth_limit=10 if s in (0, 5) else 9, basis_dim=16, shell=2.5, sdf=synth.scene_variant(s), seed=20230418 + s)
        hts.append(orc.HostTree(t.child, t.data, t.scale, t.offset, t.data_format))
        dts.append(R.N3Tree.from_arrays(t.child, t.data, t.scale, t.offset, t.data_format))
    opt = R.RenderOptions(spp=6, denoise=False)
    ctx = R.RenderContext(W, H, frames=Bc)
    seen = {}  # (scene, pose) -> RGBA8 of its first rendering
    checked_scenes = {m: set() for m in ("pose", "scene")}
    for m in ("pose", "scene"):
        for rank in range(world):
            assert set(bench.scenes_of_rank(rank, world, n_scenes, m)) == (set(range(n_scenes)) if m == "pose" else {rank})
            # this rank's frames, cut into launch groups of one scene each (bench.plan_groups); the first group of each scene it owns
            per_rank = n_scenes * n_poses // world
            groups = bench.plan_groups(per_rank, Bc, rank, world, n_poses, n_scenes, m)
            # ONE launch group per rank and map: of scene s = (rank + 3) mod 8 under the pose map (every rank holds all 8 scenes),
            # of the rank's own scene under the scene map -- in both cases the group that holds pose q = ((s - 3) mod 8) + 32,
            # so that the same (scene, pose) is rendered under both maps, by different ranks, in different slots
            s_want = (rank + 3) % n_scenes if m == "pose" else rank
            q = (s_want - 3) % n_scenes + 32
            sc, idx = next((sc, idx) for sc, idx in groups if sc == s_want and q in idx)
            if m == "pose":
                assert all(p % world == rank for p in idx)
            k = idx.index(q)
            ctx.rng_seed()
            R.launch_renderer_batch(dts[sc], [cams[p] for p in idx], opt, ctx, rng_jumps=[WARM + p for p in idx])
            ctx.select_frame(k)
            aux, rgba8 = ctx.download_aux(), ctx.download_rgba8()
            if m == "pose":  # one complete oracle frame per scene
                aux_o, rgba_o = oracle_whole_frame(hts[sc], cams[q], fx, 6, WARM + q)
                assert_bits_equal(aux, aux_o, "map %s rank %d scene %d pose %d: aux planes" % (m, rank, sc, q))
                assert np.array_equal(rgba8, orc.rgba8(rgba_o))
            else:
                assert np.array_equal(seen[(sc, q)], rgba8), "frame (%d, %d) differs between the scene maps" % (sc, q)
            seen.setdefault((sc, q), rgba8)
            checked_scenes[m].add(sc)
    assert checked_scenes["pose"] == checked_scenes["scene"] == set(range(n_scenes))
    ctx.free()
    for dt in dts:
        dt.free()
