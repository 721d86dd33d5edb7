"""HIP guided-softmax filter (rto_filtering, all levels fused in one launch) vs the oracle's
level-by-level restatement of filtering.cu:108-228: bit-exact fp32 output."""
import numpy as np
import pytest

import orc
import rt_octree_amd as R
from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _inputs(L, H, W, seed):
    rs = np.random.RandomState(seed)
    g = (rs.randn(L, H, W) * 3).astype(np.float32)
    w = rs.rand(L, H, W).astype(np.float32)
    w /= w.sum(0, keepdims=True)
    noisy = rs.rand(H, W, 4).astype(np.float32)
    noisy[..., 3] = 1
    return w, g, noisy


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("shape", [(48, 40), (33, 71), (8, 8), (3, 5)])
def test_filter_bit_exact(L, shape):
    H, W = shape
    w, g, noisy = _inputs(L, H, W, seed=L * 100 + H)
    ref = orc.filter_levels(w, g, noisy)
    dev = torch.device("cuda:0")
    tw, tg, tn = (torch.from_numpy(a).to(dev) for a in (w, g, noisy))
    out = torch.full((H, W, 4), -7.0, device=dev)
    R.filtering(torch.cuda.current_stream(), tw, tg, tn, out)
    torch.cuda.synchronize()
    assert_bits_equal(out.cpu().numpy(), ref, "filter L=%d %dx%d" % (L, H, W))


def test_filter_extreme_guidance():
    """huge logit spread: exp underflows to 0 for most taps, borders see -FLT_MAX taps."""
    L, H, W = 4, 40, 48
    w, g, noisy = _inputs(L, H, W, seed=9)
    g *= 40.0
    ref = orc.filter_levels(w, g, noisy)
    dev = torch.device("cuda:0")
    out = torch.empty((H, W, 4), device=dev)
    R.filtering(None, torch.from_numpy(w).to(dev), torch.from_numpy(g).to(dev), torch.from_numpy(noisy).to(dev), out)
    torch.cuda.synchronize()
    assert_bits_equal(out.cpu().numpy(), ref)


def test_filter_rejects_bad_levels():
    dev = torch.device("cuda:0")
    t = torch.zeros((7, 8, 8), device=dev)
    img = torch.zeros((8, 8, 4), device=dev)
    with pytest.raises(R.RtoError) as e:
        R.filtering(None, t, t, img, torch.zeros_like(img))
    assert "Kernel size == 15 not supported" in str(e.value)


def test_filter_batch_bit_exact():
    L, H, W, n = 4, 40, 56, 3
    dev = torch.device("cuda:0")
    ws, gs, ns, refs = [], [], [], []
    for i in range(n):
        w, g, noisy = _inputs(L, H, W, seed=50 + i)
        ws.append(w); gs.append(g); ns.append(noisy)
        refs.append(orc.filter_levels(w, g, noisy))
    tw, tg, tn = (torch.from_numpy(np.stack(a)).to(dev) for a in (ws, gs, ns))
    out = torch.empty((n, H, W, 4), device=dev)
    R.filtering(None, tw, tg, tn, out)
    torch.cuda.synchronize()
    for i in range(n):
        assert_bits_equal(out[i].cpu().numpy(), refs[i], "image %d" % i)


def _psnr(a, b):
    mse = float(np.mean((a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)) ** 2))
    return np.inf if mse == 0 else -10.0 * np.log10(mse)


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("shape", [(48, 40), (33, 71), (8, 8), (3, 5), (64, 64), (100, 37)])
def test_factorised_filter_matches_exact(L, shape):
    """FILTER_FAST (exp(g - tile max) once per pixel, box sums) against the bit-exact filter: the factor
    cancels, so only roundings differ -- relative 1e-5 at worst, far inside the 1e-4 dB of the float paths.
    Guidance in [0, 6] like the ReLU6 output of GuidanceNet; ragged sizes exercise the image borders."""
    H, W = shape
    rs = np.random.RandomState(L * 31 + H)
    g = rs.uniform(0, 6, (L, H, W)).astype(np.float32)
    w = rs.rand(L, H, W).astype(np.float32)
    w /= w.sum(0, keepdims=True)
    noisy = rs.rand(H, W, 4).astype(np.float32)
    ref = orc.filter_levels(w, g, noisy)
    dev = torch.device("cuda:0")
    tw, tg, tn = (torch.from_numpy(a).to(dev) for a in (w, g, noisy))
    out = torch.full((H, W, 4), -7.0, device=dev)
    R.filtering(None, tw, tg, tn, out, mode=R.FILTER_FAST)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.all(got[..., 3] == 1.0)
    assert np.allclose(got[..., :3], ref[..., :3], rtol=2e-5, atol=2e-6), np.abs(got - ref).max()
    assert _psnr(got, ref) > 110.0


def test_factorised_filter_wide_guidance_range_takes_the_safe_route():
    """a guidance range that would underflow exp(g - tile max) inside some window (here +-120): those tiles
    take the per-pixel-maximum route; others (range <= 80) stay factorised.  Also a batch of 2."""
    L, H, W = 4, 72, 90
    rs = np.random.RandomState(5)
    g = (rs.randn(2, L, H, W) * 3).astype(np.float32)
    g[0, :, :40, :50] *= 40.0   # a region of image 0 with a huge spread
    g[1] = np.clip(g[1], -30, 30)
    w = rs.rand(2, L, H, W).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    noisy = rs.rand(2, H, W, 4).astype(np.float32)
    dev = torch.device("cuda:0")
    out = torch.empty((2, H, W, 4), device=dev)
    R.filtering(None, torch.from_numpy(w).to(dev), torch.from_numpy(g).to(dev), torch.from_numpy(noisy).to(dev), out,
                mode=R.FILTER_FAST)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.all(np.isfinite(got))
    for i in range(2):
        ref = orc.filter_levels(w[i], g[i], noisy[i])
        assert np.allclose(got[i][..., :3], ref[..., :3], rtol=1e-4, atol=1e-5), (i, np.abs(got[i] - ref).max())
