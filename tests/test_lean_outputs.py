"""Lean outputs of the batched render -> denoise route (round 5, VERDICT r4 task 3; rto_ctx_set_lean_outputs).

The reference's kernel stores 48 bytes per pixel (8 aux planes + an RGBA32F image, volrend.cu:187-212) of which its denoise
stage reads 16: aux planes 0..3 -- planes 4..7 are their squares, the image's rgb is planes 0..2 again.  A lean batched launch
stores exactly those four values, interleaved, as the noisy image's (r, g, b, alpha) and nothing else.  Checked here:
the four values are the full route's aux planes 0..3 bit for bit, the aux buffer is left alone, the denoised images of both
filter routes equal the full route's bit for bit (one-call rto_denoise and the two-call form with RTO_NET_INPUT_RGBA), a
launch with denoise off or a single-frame launch keeps the full outputs."""
import numpy as np
import pytest

import rt_octree_amd as R
from helpers import assert_bits_equal
from rt_octree_amd import denoiser, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def scene():
    t = synth.make_tree(depth_limit=7, basis_dim=9, shell=2.5)
    dt = R.N3Tree.from_arrays(t.child, t.data, t.scale, t.offset, t.data_format)
    torch.manual_seed(3)
    net = denoiser.FusedGuidanceNet(denoiser.GuidanceNetCompact.from_full(denoiser.GuidanceNet(8, 32, 5, 2, 4)).eval())
    yield dt, net
    dt.free()


def cams_for(W, H, n):
    fx = synth.blender_focal(W)
    out = []
    for p in synth.orbit_poses(n):
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(p)
        out.append(c)
    return out


def views(ctx, n):
    aux, noisy, image = (torch.as_tensor(v, device="cuda:0")[:n] for v in ctx.batch_views())
    return aux, noisy, image


@pytest.mark.parametrize("W,H,bg", [(400, 304, 1.0), (333, 257, 0.25)])
def test_lean_launch_stores_planes_0_to_3_and_denoises_to_the_same_images(scene, W, H, bg):
    dt, net = scene
    n = 4
    cams = cams_for(W, H, n)
    opt = R.RenderOptions(spp=6, denoise=True, background_brightness=bg)
    jumps = [100 + i for i in range(n)]
    full = R.RenderContext(W, H, frames=n)
    full.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, full, rng_jumps=jumps)
    assert not full.frames_are_lean(0, n)
    aux_f, noisy_f, image_f = views(full, n)
    want = {}
    for mode in (R.FILTER_FAST, R.FILTER_EXACT):
        full.select_frame(0)
        net.denoise(full, n=n, mode=mode)
        torch.cuda.synchronize()
        want[mode] = image_f.clone()
    aux_full = aux_f.clone()

    lean = R.RenderContext(W, H, frames=n)
    aux_l, noisy_l, image_l = views(lean, n)
    aux_l.fill_(-3.0)  # (a lean launch must not touch the aux buffer)
    lean.set_lean_outputs(True)
    lean.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, lean, rng_jumps=jumps)
    torch.cuda.synchronize()
    assert lean.frames_are_lean(0, n) and lean.frames_are_lean(1, 2) and not lean.frames_are_lean(0, n + 1)
    assert bool(torch.all(aux_l == -3.0)), "a lean launch wrote aux planes"
    got = noisy_l.permute(0, 3, 1, 2).contiguous()  # [n][4][H][W] = (r, g, b, alpha) planes
    assert_bits_equal(got.cpu().numpy(), aux_full[:, :4].cpu().numpy(), "lean (r, g, b, alpha) vs aux planes 0..3")
    assert_bits_equal(noisy_l[..., :3].cpu().numpy(), noisy_f[..., :3].cpu().numpy(), "lean rgb vs the full noisy image")
    for mode in (R.FILTER_FAST, R.FILTER_EXACT):  # the one-call form finds the route by itself
        image_l.fill_(-7.0)
        lean.select_frame(0)
        net.denoise(lean, n=n, mode=mode)
        torch.cuda.synchronize()
        assert_bits_equal(image_l.cpu().numpy(), want[mode].cpu().numpy(), "rto_denoise on lean frames, mode %d" % mode)
    # the two-call forms with RTO_NET_INPUT_RGBA, culled and not
    marks = lean.tile_marks()
    assert marks is not None
    lean.select_frame(0)
    for cull in (None, marks):
        image_l.fill_(-7.0)
        net.forward_packed(noisy_l, rgba=True, cull=cull)
        net.filter_packed(lean.noisy_ptr, lean.image_ptr, shape=(n, H, W), cull=cull)
        torch.cuda.synchronize()
        assert_bits_equal(image_l.cpu().numpy(), want[R.FILTER_FAST].cpu().numpy(), "packed route on the lean image")
        wm, gm = net(noisy_l, rgba=True, cull=cull)
        net.filter_planes(wm, gm, lean.noisy_ptr, lean.image_ptr, mode=R.FILTER_EXACT, cull=cull)
        torch.cuda.synchronize()
        assert_bits_equal(image_l.cpu().numpy(), want[R.FILTER_EXACT].cpu().numpy(), "fp32-plane route on the lean image")
    # denoise off: the image is the final one (volrend.cu:206) -- full outputs whatever the switch says
    lean.rng_seed()
    R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=False, background_brightness=bg), lean, rng_jumps=jumps)
    torch.cuda.synchronize()
    assert not lean.frames_are_lean(0, n)
    assert_bits_equal(aux_l.cpu().numpy(), aux_full.cpu().numpy(), "denoise off: full aux planes")
    assert bool(torch.all(image_l[..., 3] == 1.0))
    # a single-frame launch into a slot a lean launch wrote: full outputs for that slot, the rest stays lean
    lean.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, lean, rng_jumps=jumps)
    lean.select_frame(1)
    lean.rng_seed()
    lean.rng_advance(jumps[1] << 32)
    R.launch_renderer(dt, cams[1], opt, lean)
    torch.cuda.synchronize()
    #  the state is per slot: slot 1 is full now, slots 0, 2, 3 still hold lean data; the mixed range is refused
    # instead of being denoised through stale aux planes, and every run of slots still denoises to the full route's images
    from rt_octree_amd._lib import lib
    assert not lean.frames_are_lean(0, n) and lib().rto_ctx_frames_are_lean(lean._h, 0, n) == -1
    assert lean.frames_are_lean(0, 1) and not lean.frames_are_lean(1, 1) and lean.frames_are_lean(2, 2)
    assert lib().rto_ctx_frames_are_lean(lean._h, 1, 1) == 0
    assert_bits_equal(aux_l[1].cpu().numpy(), aux_full[1].cpu().numpy(), "single frame after a lean launch: full aux planes")
    assert bool(torch.all(noisy_l[1, ..., 3] == 1.0))
    lean.select_frame(0)
    with pytest.raises(R.RtoError):
        net.denoise(lean, n=n, mode=R.FILTER_EXACT)
    for mode in (R.FILTER_FAST, R.FILTER_EXACT):
        image_l.fill_(-7.0)
        for first, cnt in ((0, 1), (1, 1), (2, 2)):
            lean.select_frame(first)
            net.denoise(lean, n=cnt, mode=mode)
        torch.cuda.synchronize()
        assert_bits_equal(image_l.cpu().numpy(), want[mode].cpu().numpy(), "runs of lean / full / lean slots, mode %d" % mode)
    lean.select_frame(0)


@pytest.mark.parametrize("W,H,bg", [(400, 304, 1.0), (333, 257, 0.25)])
def test_sparse_lean_launch_denoises_to_the_same_images(scene, W, H, bg):
    """round 6, rto_ctx_set_lean_outputs level 2: as lean, and NOTHING is stored for the pixels of the tiles the launch's culling
    left unmarked -- the noisy image is poisoned beforehand and the poison must survive there -- yet the factorised route
    (one-call rto_denoise; two-call form with RTO_NET_INPUT_SPARSE + the marks) denoises to the full route's images bit for bit:
    the network takes the background for those pixels and stores no maps for the tiles it skips, the filter substitutes both.
    The noisy image comes back complete from rto_ctx_download_image / _rgba8; the bit-exact filter route refuses sparse frames."""
    dt, net = scene
    n = 4
    cams = cams_for(W, H, n)
    opt = R.RenderOptions(spp=6, denoise=True, background_brightness=bg)
    jumps = [100 + i for i in range(n)]
    full = R.RenderContext(W, H, frames=n)
    full.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, full, rng_jumps=jumps)
    aux_f, noisy_f, image_f = views(full, n)
    full.select_frame(0)
    net.denoise(full, n=n, mode=R.FILTER_FAST)
    torch.cuda.synchronize()
    want = image_f.clone()

    sp = R.RenderContext(W, H, frames=n)
    aux_s, noisy_s, image_s = views(sp, n)
    POISON = -12345.0
    noisy_s.fill_(POISON)
    sp.set_lean_outputs(2)
    sp.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, sp, rng_jumps=jumps)
    torch.cuda.synchronize()
    assert sp.frames_lean_level(0, n) == 2 and sp.frames_are_lean(0, n)
    marks = sp.tile_marks()
    assert marks is not None
    for f in range(n):
        sp.select_frame(f)
        patched = sp.download_image(noisy=True)  # completed on the host
        raw = noisy_s[f].cpu().numpy()
        unm = raw[..., 0] == POISON
        assert np.all(raw[unm] == POISON), "a sparse launch stored only part of a culled pixel"
        assert unm.mean() > 0.2, "the test scene must have culled tiles"
        # unmarked pixels come in whole 8x8 tiles and are the background after the host-side completion
        for (y, x) in [(0, 0), (H - 1, W - 1)]:
            assert unm[y, x] == unm[(y // 8) * 8, (x // 8) * 8]
        assert np.all(patched[unm][:, :3] == np.float32(bg)) and np.all(patched[unm][:, 3] == 0.0)
        # ... and everything equals the lean / full route's values: (r, g, b) = the full noisy image, alpha = aux plane 3
        assert_bits_equal(patched[..., :3], noisy_f[f, ..., :3].cpu().numpy(), "sparse noisy image (completed) vs the full route")
        assert_bits_equal(patched[..., 3], aux_f[f, 3].cpu().numpy(), "sparse alpha vs aux plane 3")
        r8 = sp.download_rgba8(noisy=True)
        assert np.array_equal(r8[..., :3], (patched[..., :3] * 255).astype(np.uint8))
    sp.select_frame(0)
    image_s.fill_(-7.0)
    net.denoise(sp, n=n, mode=R.FILTER_FAST)  # picks the sparse route by itself
    torch.cuda.synchronize()
    assert_bits_equal(image_s.cpu().numpy(), want.cpu().numpy(), "rto_denoise on sparse lean frames")
    with pytest.raises(R.RtoError):
        net.denoise(sp, n=n, mode=R.FILTER_EXACT)
    image_s.fill_(-7.0)
    net.forward_packed(noisy_s, rgba=True, sparse=True, cull=marks)
    net.filter_packed(sp.noisy_ptr, sp.image_ptr, shape=(n, H, W), cull=marks)
    torch.cuda.synchronize()
    assert_bits_equal(image_s.cpu().numpy(), want.cpu().numpy(), "two-call form with RTO_NET_INPUT_SPARSE")
    with pytest.raises(R.RtoError):  # sparse maps without the marks: refused
        net.filter_packed(sp.noisy_ptr, sp.image_ptr, shape=(n, H, W))
    # a later launch into the context replaces the marks: the sparse frames can then no longer be completed or denoised
    sp.set_lean_outputs(0)
    one = R.RenderContext(W, H)
    sp.select_frame(0)
    sp.rng_seed()
    R.launch_renderer(dt, cams[0], opt, sp)  # slot 0 becomes a full frame, the marks of the batch are gone
    torch.cuda.synchronize()
    assert sp.frames_lean_level(0, 1) == 0 and sp.frames_lean_level(1, 3) == 2 and sp.frames_lean_level(0, n) == -1
    sp.select_frame(1)
    with pytest.raises(R.RtoError):
        sp.download_image(noisy=True)
    with pytest.raises(R.RtoError):
        net.denoise(sp, n=3, mode=R.FILTER_FAST)
    one.free()


@pytest.mark.parametrize("seed", range(10))
def test_sparse_lean_equals_lean_on_random_sizes_and_poses(scene, seed):
    """the sparse route's tile arithmetic (render tiles 8x8, network tiles 32x8 + halo 2, filter tiles 32x16 + halo 4 + 2) on frame
    sizes that are no multiple of anything, narrower than a strip, with cameras close to / far from / looking past the model (whole
    frames kept or culled) and several backgrounds: denoised images and the completed noisy image equal the lean route's bit for bit"""
    dt, net = scene
    rs = np.random.RandomState(1000 + seed)
    W = int(rs.choice([24, 40, 71, 97, 130, 161, 200, 257, 320, 403]))
    H = int(rs.choice([17, 33, 48, 75, 96, 131, 180, 241]))
    n = int(rs.randint(1, 5))
    bg = float(rs.choice([0.0, 0.25, 1.0]))
    fx = synth.blender_focal(W) * float(rs.uniform(0.5, 2.0))
    cams = []
    for _ in range(n):
        kind = rs.randint(0, 4)
        if kind == 0:  # the usual orbit
            pose = synth.orbit_poses(16)[rs.randint(0, 16)]
        elif kind == 1:  # close to the model: culling keeps the whole frame or most of it
            pose = synth.look_at_c2w(tuple(rs.uniform(-1.2, 1.2, 3)), target=tuple(rs.uniform(-0.3, 0.3, 3)))
        elif kind == 2:  # far away: a few marked tiles in the middle
            pose = synth.look_at_c2w(tuple(rs.uniform(6, 9, 3) * rs.choice([-1, 1], 3)), target=(0.0, 0.0, 0.0))
        else:  # looking past it: the model in a corner or out of view
            pose = synth.look_at_c2w(tuple(rs.uniform(2.5, 4, 3)), target=tuple(rs.uniform(-3, 3, 3)))
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(pose)
        cams.append(c)
    opt = R.RenderOptions(spp=6, denoise=True, background_brightness=bg)
    jumps = [int(j) for j in rs.randint(0, 500, n)]
    out = {}
    for level in (1, 2):
        ctx = R.RenderContext(W, H, frames=n)
        aux_v, noisy_v, image_v = views(ctx, n)
        noisy_v.fill_(float("nan"))
        image_v.fill_(-7.0)
        ctx.set_lean_outputs(level)
        ctx.rng_seed()
        R.launch_renderer_batch(dt, cams, opt, ctx, rng_jumps=jumps)
        noisy = []
        for f in range(n):
            ctx.select_frame(f)
            noisy.append(ctx.download_image(noisy=True))
        ctx.select_frame(0)
        net.denoise(ctx, n=n, mode=R.FILTER_FAST)
        torch.cuda.synchronize()
        out[level] = (np.stack(noisy), image_v.cpu().numpy().copy())
        ctx.free()
    assert_bits_equal(out[2][0], out[1][0], "noisy image (completed) %dx%d n %d bg %g" % (W, H, n, bg))
    assert_bits_equal(out[2][1], out[1][1], "denoised image %dx%d n %d bg %g" % (W, H, n, bg))
