"""The denoise stage's share of the empty-space culling (rto_filtering_packed_culled): filter workgroups whose inputs
all lie in culled render tiles copy the filter's background tile instead of filtering.  The copied values are the ones the
same kernel computes on a synthetic background frame, so the output must equal rto_filtering_packed's bit for bit -- on full
frames, for white and for grey backgrounds, with culling off (every tile marked) and after a single-frame launch (no marks)."""
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


def cams_for(W, H, n, radius=4.0311):
    fx = synth.blender_focal(W)
    out = []
    for p in synth.orbit_poses(n, radius=radius):
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(p)
        out.append(c)
    return out


def images(ctx, n):
    return torch.as_tensor(ctx.batch_views()[2], device="cuda:0")[:n].clone()


@pytest.mark.parametrize("W,H,bg,radius", [(800, 800, 1.0, 4.0311), (400, 304, 0.3, 4.0311), (333, 257, 0.0, 6.0), (160, 120, 1.0, 4.0311)])
def test_culled_filter_is_bit_identical(scene, W, H, bg, radius):
    dt, net = scene
    n = 5
    cams = cams_for(W, H, n, radius)
    opt = R.RenderOptions(spp=6, denoise=True, background_brightness=bg)
    ctx = R.RenderContext(W, H, frames=n)
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, ctx)
    marks = ctx.tile_marks()
    assert marks is not None and marks[3] == n and marks[2] == 0 and marks[4] == np.float32(bg)
    live, total = ctx.queue_stats()
    assert 0 < live < total
    aux = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
    net.forward_packed(aux[:n], squares_implied=True)
    ctx.select_frame(0)
    net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
    torch.cuda.synchronize()
    plain = images(ctx, n)
    torch.as_tensor(ctx.batch_views()[2], device="cuda:0").fill_(-7.0)
    net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W), cull=marks)
    torch.cuda.synchronize()
    culled = images(ctx, n)
    assert_bits_equal(culled.cpu().numpy(), plain.cpu().numpy(), "culled filter vs plain filter, %dx%d bg %g" % (W, H, bg))
    # the network's share: maps computed with tile skipping, then the plain filter on them
    net.forward_packed(aux[:n], squares_implied=True, cull=marks)
    torch.as_tensor(ctx.batch_views()[2], device="cuda:0").fill_(-7.0)
    net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n).cpu().numpy(), plain.cpu().numpy(), "culled network vs plain network, %dx%d bg %g" % (W, H, bg))
    # ... and without the squares shortcut, both stages culled
    net.forward_packed(aux[:n], squares_implied=False, cull=marks)
    net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W), cull=marks)
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n).cpu().numpy(), plain.cpu().numpy(), "both stages culled, %dx%d bg %g" % (W, H, bg))
    net.forward_packed(aux[:n], squares_implied=True)
    # the fp32-plane routes (the reference's tensors): maps with tile skipping == maps without; exact and factorised filters
    wm, gm = (t.clone() for t in net(aux[:n], squares_implied=True))
    wm_c, gm_c = net(aux[:n], squares_implied=True, cull=marks)
    torch.cuda.synchronize()
    assert_bits_equal(wm_c.cpu().numpy(), wm.cpu().numpy(), "weight planes, culled network")
    assert_bits_equal(gm_c.cpu().numpy(), gm.cpu().numpy(), "guidance planes, culled network")
    R.filtering(None, wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_EXACT)
    torch.cuda.synchronize()
    plain_exact = images(ctx, n)
    torch.as_tensor(ctx.batch_views()[2], device="cuda:0").fill_(-7.0)
    net.filter_planes(wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_EXACT, cull=marks)
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n).cpu().numpy(), plain_exact.cpu().numpy(), "exact filter, culled")
    torch.as_tensor(ctx.batch_views()[2], device="cuda:0").fill_(-7.0)
    net.filter_planes(wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_FAST, cull=marks)
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n).cpu().numpy(), plain.cpu().numpy(), "factorised filter on fp32 planes, culled")
    if W == 800:
        # the frames do hold workgroups that were copied (else this test proves nothing): workgroup (1, 1) of frame 0 sees only
        # sky -- a poisoned input pixel inside it changes what the plain filter writes there, and not what the culled one does
        noisy = torch.as_tensor(ctx.batch_views()[1], device="cuda:0")
        assert bool(torch.all(aux[0, 3, 24:72, 24:72] == 0))
        noisy[0, 48, 48, :3] = 0.5
        net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W), cull=marks)
        torch.cuda.synchronize()
        assert torch.equal(images(ctx, n), plain)
        net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
        torch.cuda.synchronize()
        assert not torch.equal(images(ctx, n)[0, 40:56, 40:56], plain[0, 40:56, 40:56])
        net.filter_planes(wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_EXACT, cull=marks)
        torch.cuda.synchronize()
        assert torch.equal(images(ctx, n), plain_exact)  # (the exact filter skipped that workgroup as well)
        # the same for the network: a poisoned aux value inside a skipped network tile does not reach the maps
        noisy[0, 48, 48, :3] = bg
        aux[0, 0, 48, 48] = 0.5
        net.forward_packed(aux[:n], squares_implied=True, cull=marks)
        net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
        torch.cuda.synchronize()
        assert torch.equal(images(ctx, n), plain)
        net.forward_packed(aux[:n], squares_implied=True)
        net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
        torch.cuda.synchronize()
        assert not torch.equal(images(ctx, n)[0, 40:56, 40:56], plain[0, 40:56, 40:56])
    ctx.free()


def test_no_marks_no_skip(scene):
    """culling off: every tile counts as marked, nothing is copied; a single-frame launch leaves no marks at all"""
    dt, net = scene
    W, H, n = 320, 240, 3
    cams = cams_for(W, H, n)
    opt = R.RenderOptions(spp=6, denoise=True)
    ctx = R.RenderContext(W, H, frames=n)
    ctx.set_tuning("cull", 0)
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, ctx)
    marks = ctx.tile_marks()
    aux = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
    net.forward_packed(aux[:n], squares_implied=True)
    ctx.select_frame(0)
    # a poisoned input image: if any workgroup were copied instead of filtered the poison would not show in its output
    noisy = torch.as_tensor(ctx.batch_views()[1], device="cuda:0")
    noisy[:n, :, :, :3] = 0.25
    net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
    torch.cuda.synchronize()
    plain = images(ctx, n)
    net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W), cull=marks)
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n).cpu().numpy(), plain.cpu().numpy(), "culling off")
    R.launch_renderer(dt, cams[0], opt, ctx)
    assert ctx.tile_marks() is None  # (culling off on this context: the single-frame kernel marks nothing either)
    ctx.free()


def test_refuses_foreign_marks(scene):
    dt, net = scene
    W, H, n = 160, 120, 2
    ctx = R.RenderContext(W, H, frames=n)
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams_for(W, H, n), R.RenderOptions(spp=6, denoise=True), ctx)
    aux = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
    net.forward_packed(aux[:n], squares_implied=True)
    ctx.select_frame(0)
    p, words, s0, frames, bg = ctx.tile_marks()
    with pytest.raises(R.RtoError, match="tile marks"):
        net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W), cull=(p, words + 1, s0, frames, bg))
    with pytest.raises(ValueError):
        net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W), cull=(p, words, s0, 1, bg))
    ctx.free()


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_one_call_denoise_equals_the_separate_calls(scene, mode):
    """rto_denoise (Denoiser::denoise in one call) == network + filter called separately, after a batched launch (tile
    marks used) and after a single-frame launch into a later slot (no marks)"""
    dt, net = scene
    W, H, n = 400, 304, 4
    cams = cams_for(W, H, n)
    opt = R.RenderOptions(spp=6, denoise=True)
    ctx = R.RenderContext(W, H, frames=n)
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, ctx)
    aux = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
    fmode = R.FILTER_FAST if mode == "fast" else R.FILTER_EXACT
    ctx.select_frame(0)
    if mode == "fast":
        net.forward_packed(aux[:n], squares_implied=True)
        net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
    else:
        wm, gm = net(aux[:n], squares_implied=True)
        R.filtering(None, wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_EXACT)
    torch.cuda.synchronize()
    want = images(ctx, n)
    torch.as_tensor(ctx.batch_views()[2], device="cuda:0").fill_(-7.0)
    net.denoise(ctx, n, fmode)
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n).cpu().numpy(), want.cpu().numpy(), "rto_denoise after a batched launch, " + mode)
    # a single frame into slot 2, with its own marks (round 4: the single-frame kernel culls too when asked), same pixels as
    # slot 2 of the batch (same camera, same RNG jump)
    ctx.select_frame(2)
    ctx.rng_seed()
    ctx.rng_advance(2 << 32)
    R.launch_renderer(dt, cams[2], opt, ctx)
    assert ctx.tile_marks() is None  # (default: the lone frame does not cull)
    ctx.set_tuning("cull_single", 1)
    R.launch_renderer(dt, cams[2], opt, ctx)
    assert ctx.tile_marks()[2:4] == (2, 1)  # first slot 2, one frame
    torch.as_tensor(ctx.batch_views()[2], device="cuda:0")[2].fill_(-7.0)
    net.denoise(ctx, 1, fmode)
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n)[2].cpu().numpy(), want[2].cpu().numpy(), "rto_denoise after a single-frame launch, " + mode)
    with pytest.raises(R.RtoError, match="frames from slot"):
        net.denoise(ctx, 3, fmode)  # slot 2 + 3 frames > 4
    ctx.free()


def _cull_seeds():
    import os
    spec = os.environ.get("RTO_FUZZ_SEEDS", "")
    if ":" in spec:
        a, b = spec.split(":")
        return range(int(a), int(b))
    return range(12)


@pytest.mark.parametrize("seed", _cull_seeds())
def test_random_frames_culled_denoise_is_bit_identical(scene, seed):
    """random frame sizes, backgrounds, batch sizes and cameras (orbiting, far, inside the volume, looking past the model):
    both denoise routes with the tile marks == without, bit for bit"""
    dt, net = scene
    rs = np.random.RandomState(500 + seed)
    W, H = int(rs.randint(40, 520)), int(rs.randint(40, 420))
    n = int(rs.randint(1, 5))
    fx = float(rs.uniform(0.6, 2.5) * W)
    cams = []
    for _ in range(n):
        mode = rs.rand()
        if mode < 0.15:
            pos = rs.uniform(-0.6, 0.6, 3)
        elif mode < 0.4:
            pos = rs.randn(3)
            pos = pos / np.linalg.norm(pos) * rs.uniform(7, 15)
        else:
            pos = rs.randn(3)
            pos = pos / np.linalg.norm(pos) * rs.uniform(2.5, 5.0)
        target = rs.uniform(-0.4, 0.4, 3) if rs.rand() < 0.75 else rs.uniform(-3, 3, 3)
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(synth.look_at_c2w(pos, target))
        cams.append(c)
    bg = float(rs.choice([1.0, 0.0, rs.rand()]))
    opt = R.RenderOptions(spp=int(rs.choice([1, 4, 6])), denoise=True, background_brightness=bg)
    ctx = R.RenderContext(W, H, frames=n)
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams, opt, ctx, rng_jumps=[int(rs.randint(0, 300)) for _ in range(n)])
    marks = ctx.tile_marks()
    aux = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
    ctx.select_frame(0)
    net.forward_packed(aux[:n], squares_implied=True)
    net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
    torch.cuda.synchronize()
    plain = images(ctx, n)
    wm, gm = (t.clone() for t in net(aux[:n], squares_implied=True))
    R.filtering(None, wm, gm, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_EXACT)
    torch.cuda.synchronize()
    plain_exact = images(ctx, n)
    net.denoise(ctx, n, R.FILTER_FAST)
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n).cpu().numpy(), plain.cpu().numpy(), "seed %d packed route %dx%d n %d bg %g" % (seed, W, H, n, bg))
    net.denoise(ctx, n, R.FILTER_EXACT)
    torch.cuda.synchronize()
    assert_bits_equal(images(ctx, n).cpu().numpy(), plain_exact.cpu().numpy(), "seed %d exact route %dx%d n %d bg %g" % (seed, W, H, n, bg))
    wm_c, gm_c = net(aux[:n], squares_implied=True, cull=marks)
    torch.cuda.synchronize()
    assert torch.equal(wm_c, wm) and torch.equal(gm_c, gm)
    ctx.free()
