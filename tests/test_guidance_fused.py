"""Fused GuidanceNet kernel (MFMA fp16, fp32 accumulate) vs the fp32 PyTorch compact network.
Tolerance: fp16 rounding of inputs / weights / activations -- |guidance| <= 3e-2 absolute (values in
[0, 6]), softmax weights <= 1e-2; and the denoised image through the HIP filter stays > 50 dB PSNR
from the one obtained with the PyTorch network (SURVEY.md 8c tolerance for the fp16 network)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import denoiser  # noqa: E402

pytestmark = pytest.mark.gpu


def _nets(seed=0):
    torch.manual_seed(seed)
    full = denoiser.GuidanceNet(8, 32, 5, 2, 4)
    compact = denoiser.GuidanceNetCompact.from_full(full).eval()
    return compact, denoiser.FusedGuidanceNet(compact)


@pytest.mark.parametrize("shape", [(1, 64, 96), (2, 37, 53), (1, 8, 32), (3, 5, 7), (1, 20, 333), (2, 9, 161)])  # 333: strips of 5 + 5 + 1 tiles
def test_fused_matches_fp32_network(shape):
    n, H, W = shape
    compact, fused = _nets()
    torch.manual_seed(1)
    aux = torch.rand(n, 8, H, W)
    aux[:, 4:] = aux[:, :4] ** 2
    with torch.no_grad():
        w_ref, g_ref = compact(aux)
    w, g = fused(aux.cuda().contiguous())
    torch.cuda.synchronize()
    assert float((g.cpu() - g_ref).abs().max()) < 3e-2
    assert float((w.cpu() - w_ref).abs().max()) < 1e-2
    assert np.allclose(w.cpu().sum(1).numpy(), 1.0, atol=1e-5)


def test_fused_exact_on_integer_data():
    """layout check (MFMA fragment maps, tap order, halo): small-integer weights and inputs make every
    product and sum exact in fp16/fp32, so the fused kernel must equal the fp32 network exactly,
    asymmetric kernels included."""
    compact, _ = _nets()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for layer in compact.layers:
            layer.conv.weight.copy_(torch.randint(-2, 3, layer.conv.weight.shape, generator=g).float() / 8)
            layer.conv.bias.copy_(torch.randint(-4, 5, layer.conv.bias.shape, generator=g).float() / 8)
    fused = denoiser.FusedGuidanceNet(compact)
    aux = torch.randint(0, 3, (1, 8, 24, 40), generator=g).float() / 4
    with torch.no_grad():
        w_ref, g_ref = compact(aux)
    w, gm = fused(aux.cuda().contiguous())
    torch.cuda.synchronize()
    assert torch.equal(gm.cpu(), g_ref)
    assert float((w.cpu() - w_ref).abs().max()) < 2e-6  # softmax: fast exp vs torch exp


def test_denoised_image_psnr():
    compact, fused = _nets()
    torch.manual_seed(2)
    H, W = 96, 128
    aux = torch.rand(1, 8, H, W)
    noisy = torch.rand(H, W, 4)
    noisy[..., 3] = 1
    dev = torch.device("cuda:0")
    with torch.no_grad():
        w_ref, g_ref = compact(aux)
    w, g = fused(aux.to(dev).contiguous())
    outs = []
    for wm, gm in ((w_ref[0].to(dev).contiguous(), g_ref[0].to(dev).contiguous()), (w[0], g[0])):
        out = torch.empty((H, W, 4), device=dev)
        R.filtering(None, wm, gm, noisy.to(dev), out)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    mse = np.mean((outs[0][..., :3].astype(np.float64) - outs[1][..., :3]) ** 2)
    assert -10 * np.log10(mse) > 50.0


@pytest.mark.parametrize("shape", [(2, 37, 53), (1, 64, 96), (1, 17, 350)])
def test_squares_implied_reads_half_the_input_bit_identically(shape):
    """RTO_NET_AUX_SQUARES_IMPLIED: the renderer's aux planes 4..7 are the fp32 squares of planes 0..3; the kernel
    may square them itself instead of reading them -- same fp32 products, same fp16 inputs, same maps."""
    n, H, W = shape
    _, fused = _nets(3)
    torch.manual_seed(4)
    aux = torch.rand(n, 8, H, W)
    aux[:, 4:] = aux[:, :4] * aux[:, :4]
    dev_aux = aux.cuda().contiguous()
    w0, g0 = (t.clone() for t in fused(dev_aux))
    poisoned = dev_aux.clone()
    poisoned[:, 4:] = 123.0  # must not be read in the implied mode
    w1, g1 = fused(poisoned, squares_implied=True)
    torch.cuda.synchronize()
    assert torch.equal(w0, w1) and torch.equal(g0, g1)


@pytest.mark.parametrize("shape", [(2, 37, 53), (1, 64, 96), (3, 100, 41), (1, 40, 330)])
def test_packed_denoise_route_equals_the_fp32_maps_route(shape):
    """rto_guidance_net_forward_packed + rto_filtering_packed (fp16 logits + guidance, 16 B per pixel, softmax taken
    by the filter) == rto_guidance_net_forward + the factorised filter on fp32 maps, bit for bit."""
    n, H, W = shape
    _, fused = _nets(5)
    torch.manual_seed(6)
    aux = torch.rand(n, 8, H, W)
    aux[:, 4:] = aux[:, :4] * aux[:, :4]
    dev = torch.device("cuda:0")
    aux_d = aux.to(dev).contiguous()
    noisy = torch.rand(n, H, W, 4, device=dev)
    w, g = fused(aux_d, squares_implied=True)
    ref = torch.empty_like(noisy)
    R.filtering(None, w, g, noisy, ref, mode=R.FILTER_FAST)
    out = torch.full_like(noisy, -3.0)
    fused.forward_packed(aux_d, squares_implied=True)
    fused.filter_packed(noisy, out)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_packed_filter_refuses_images_of_another_extent():
    """rto_filtering_packed is handed bare image pointers: it is told their extent and refuses one that differs from the
    maps' -- a handle shared between contexts of different batch sizes cannot overrun the smaller one"""
    _, fused = _nets(5)
    dev = torch.device("cuda:0")
    aux = torch.rand(3, 8, 40, 56, device=dev)
    fused.reserve(3, 40, 56)  # the scratch at its final size: no allocation on the first forward
    fused.forward_packed(aux)
    noisy = torch.rand(3, 40, 56, 4, device=dev)
    out = torch.empty_like(noisy)
    fused.filter_packed(noisy, out, shape=(3, 40, 56))
    for bad in ((1, 40, 56), (3, 56, 40), (4, 40, 56)):
        with pytest.raises(R.RtoError) as e:
            fused.filter_packed(noisy, out, shape=bad)
        assert "packed maps hold 3 x 40 x 56" in str(e.value)
    with pytest.raises(R.RtoError):  # host memory is not an image on the network's device
        import numpy as np
        host = np.zeros((3, 40, 56, 4), np.float32)
        fused.filter_packed(noisy, host.ctypes.data, shape=(3, 40, 56))
    # a smaller forward re-labels the maps: the old extent is now the wrong one
    fused.forward_packed(aux[:1].contiguous())
    with pytest.raises(R.RtoError):
        fused.filter_packed(noisy, out, shape=(3, 40, 56))
    fused.filter_packed(noisy[:1], out[:1], shape=(1, 40, 56))
    torch.cuda.synchronize()


def _emulate_fp16_network(compact, aux):
    """The reference's half pipeline (network.py:104-118 on a `.half()` module) in float64 with roundings at ITS points:
    fp16 input and weights, exact products and sums, + bias, ReLU6, round to fp16 after each layer.  What remains
    between this and the kernel is the fp32 accumulation order inside the MFMA, i.e. at most one fp16 ulp after the
    final rounding."""
    import torch.nn.functional as F
    x = aux.half().double()
    for layer in compact.layers:
        w = layer.conv.weight.detach().half().double()
        b = layer.conv.bias.detach().half().double()
        x = F.conv2d(x, w, b, padding=1).clamp(0.0, 6.0).half().double()
    L = x.shape[1] // 2
    return torch.softmax(x[:, :L].float(), 1), x[:, L:].float()


@pytest.mark.parametrize("shape", [(1, 48, 64), (2, 33, 47)])
def test_fused_against_the_half_pipeline_rounding_points(shape):
    """tighter than the fp32-network comparison: against the reference's fp16 rounding points.  The fp32 accumulation
    order inside the MFMA can flip the fp16 rounding of a layer-1 activation by one ulp (<= 2^-8 below 4, 2^-7 above),
    which layer 2 passes on scaled by a weight: the guidance map is within ONE fp16 ulp of the emulation for > 92 % of
    the values and within 4e-3 everywhere (the fp32-network test allows 3e-2), the softmax weights within 4e-3."""
    n, H, W = shape
    compact, fused = _nets(7)
    torch.manual_seed(8)
    aux = torch.rand(n, 8, H, W)
    aux[:, 4:] = aux[:, :4] ** 2
    w_ref, g_ref = _emulate_fp16_network(compact, aux)
    w, g = fused(aux.cuda().contiguous())
    torch.cuda.synchronize()
    g, w = g.cpu(), w.cpu()
    ulp = torch.maximum(torch.tensor(2.0 ** -24), 2.0 ** (torch.floor(torch.log2(g_ref.clamp_min(2.0 ** -14))) - 10))
    diff = (g - g_ref).abs()
    assert float(diff.max()) < 4e-3
    assert float((diff > ulp * 1.0001 + 3e-5).float().mean()) < 0.08
    assert float((diff > 0).float().mean()) < 0.25
    assert float((w - w_ref).abs().max()) < 4e-3
