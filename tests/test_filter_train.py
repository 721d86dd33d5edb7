"""Training side of the guided filter (filtering.cu:230-301, 596-707).

CPU: the oracle's forward-with-saves and backward against autograd through an independent float64
PyTorch statement of the filter (unfold + softmax).  GPU: the HIP kernels against the oracle, bit for
bit, and the torch.autograd.Function wrapper (`denoiser.filtering_autograd`, the reference's
`_denoiser.filtering_autograd`) against the same float64 autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import orc


def torch_filter(weight, guidance, img):
    """weight, guidance [L,H,W]; img [H,W,4] -> [H,W,4]; float64, differentiable"""
    L, H, W = guidance.shape
    rgb = img[..., :3].permute(2, 0, 1)[None]  # [1,3,H,W]
    out = torch.zeros(3, H * W, dtype=img.dtype)
    for l in range(L):
        S = l + 1
        K = 2 * S + 1
        g = F.pad(guidance[l][None, None], (S, S, S, S), value=float("-inf"))
        g = F.unfold(g, K)[0]  # [K*K, H*W]
        k = torch.softmax(g, dim=0)
        t = F.unfold(F.pad(rgb, (S, S, S, S)), K)[0].reshape(3, K * K, H * W)
        out = out + weight[l].reshape(1, -1) * (k[None] * t).sum(1)
    res = torch.cat([out.reshape(3, H, W).permute(1, 2, 0), torch.ones(H, W, 1, dtype=img.dtype)], -1)
    return res


def _inputs(L, H, W, seed):
    rs = np.random.RandomState(seed)
    weight = rs.rand(L, H, W).astype(np.float32)
    weight /= weight.sum(0, keepdims=True)
    guidance = (rs.rand(L, H, W) * 6).astype(np.float32)  # relu6 range of the network
    img = rs.rand(H, W, 4).astype(np.float32)
    img[..., 3] = 1
    grad_out = rs.randn(H, W, 4).astype(np.float32)
    return weight, guidance, img, grad_out


def _torch_grads(weight, guidance, img, grad_out):
    w = torch.tensor(weight, dtype=torch.float64, requires_grad=True)
    g = torch.tensor(guidance, dtype=torch.float64, requires_grad=True)
    out = torch_filter(w, g, torch.tensor(img, dtype=torch.float64))
    out.backward(torch.tensor(grad_out, dtype=torch.float64))
    return out.detach().numpy(), w.grad.numpy(), g.grad.numpy()


@pytest.mark.parametrize("L,H,W", [(4, 20, 28), (2, 9, 7), (1, 5, 6), (6, 16, 16)])
def test_oracle_forward_saves_and_backward_vs_float64_autograd(L, H, W):
    weight, guidance, img, grad_out = _inputs(L, H, W, 1 + L)
    out, rf, mx, inv = orc.filter_train_forward(weight, guidance, img)
    assert np.array_equal(out.view(np.uint32), orc.filter_levels(weight, guidance, img).view(np.uint32))
    ref_out, ref_gw, ref_gg = _torch_grads(weight, guidance, img, grad_out)
    assert np.allclose(out, ref_out, rtol=2e-5, atol=2e-6)
    # saved tensors: window max, 1 / sum exp(g - max), softmax-filtered rgb
    for l in range(L):
        S = l + 1
        gp = np.pad(guidance[l], S, constant_values=-np.inf)
        win = np.lib.stride_tricks.sliding_window_view(gp, (2 * S + 1, 2 * S + 1)).reshape(H, W, -1)
        assert np.array_equal(mx[l], win.max(-1))
        assert np.allclose(inv[l], 1.0 / np.exp(win.astype(np.float64) - win.max(-1, keepdims=True)).sum(-1), rtol=1e-5)
    assert np.all(rf[..., 3] == 0)
    gw, gg = orc.filter_backward(grad_out, img, weight, guidance, rf, mx, inv)
    assert np.allclose(gw, ref_gw, rtol=1e-4, atol=1e-5)
    assert np.allclose(gg, ref_gg, rtol=1e-3, atol=2e-5)


def test_oracle_backward_rejects_bad_levels():
    z = np.zeros((7, 4, 4), np.float32)
    with pytest.raises(RuntimeError):
        orc.filter_backward(np.zeros((4, 4, 4)), np.zeros((4, 4, 4)), z, z, np.zeros((7, 4, 4, 4)), z, z)


# ----------------------------------------------------------------------------- GPU: HIP kernels
def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("L,H,W", [(4, 37, 70), (1, 8, 33), (6, 19, 40), (3, 64, 64)])
def test_hip_train_forward_and_backward_equal_oracle(L, H, W):
    """bit for bit: output, the three saved tensors, grad_weight, grad_guidance"""
    import rt_octree_amd as R
    from rt_octree_amd._lib import check, lib
    weight, guidance, img, grad_out = _inputs(L, H, W, 10 + L)
    out_o, rf_o, mx_o, inv_o = orc.filter_train_forward(weight, guidance, img)
    gw_o, gg_o = orc.filter_backward(grad_out, img, weight, guidance, rf_o, mx_o, inv_o)
    w, g, x, go = _dev(weight[None]), _dev(guidance[None]), _dev(img[None]), _dev(grad_out[None])
    out = torch.empty_like(x)
    rf = torch.empty((1, L, H, W, 4), device="cuda:0")
    mx, inv = torch.empty((1, L, H, W), device="cuda:0"), torch.empty((1, L, H, W), device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    check(lib().rto_filtering_train_forward(s, w.data_ptr(), g.data_ptr(), L, H, W, 1, x.data_ptr(), out.data_ptr(),
                                            rf.data_ptr(), mx.data_ptr(), inv.data_ptr()))
    gw, gg = torch.empty_like(w), torch.empty_like(g)
    check(lib().rto_filtering_backward(s, go.data_ptr(), x.data_ptr(), w.data_ptr(), g.data_ptr(), rf.data_ptr(),
                                       mx.data_ptr(), inv.data_ptr(), L, H, W, 1, gw.data_ptr(), gg.data_ptr()))
    torch.cuda.synchronize()
    bits = lambda t: t.cpu().numpy().view(np.uint32)
    assert np.array_equal(bits(out[0]), out_o.view(np.uint32))
    assert np.array_equal(bits(rf[0]), rf_o.view(np.uint32))
    assert np.array_equal(bits(mx[0]), mx_o.view(np.uint32))
    assert np.array_equal(bits(inv[0]), inv_o.view(np.uint32))
    assert np.array_equal(bits(gw[0]), gw_o.view(np.uint32))
    assert np.array_equal(bits(gg[0]), gg_o.view(np.uint32))


@pytest.mark.gpu
def test_autograd_function_matches_float64_autograd_and_is_deterministic():
    """denoiser.filtering_autograd (= _denoiser.filtering_autograd) on a batch of 2, against autograd
    through the float64 PyTorch statement; two backward passes give identical bits (no atomics)."""
    from rt_octree_amd import denoiser
    L, H, W, B = 4, 24, 40, 2
    ws, gs, xs, gos = zip(*[_inputs(L, H, W, 30 + b) for b in range(B)])
    w = _dev(np.stack(ws)).requires_grad_(True)
    g = _dev(np.stack(gs)).requires_grad_(True)
    x, go = _dev(np.stack(xs)), _dev(np.stack(gos))
    out = denoiser.filtering_autograd(w, g, x, requires_grad=True)
    out.backward(go)
    gw1, gg1 = w.grad.clone(), g.grad.clone()
    w.grad = g.grad = None
    denoiser.filtering_autograd(w, g, x, requires_grad=True).backward(go)
    assert torch.equal(gw1, w.grad) and torch.equal(gg1, g.grad)
    assert torch.equal(out, denoiser.filtering_autograd(w, g, x))  # inference path: same image
    for b in range(B):
        ref_out, ref_gw, ref_gg = _torch_grads(ws[b], gs[b], xs[b], gos[b])
        assert np.allclose(out[b].detach().cpu().numpy(), ref_out, rtol=2e-5, atol=2e-6)
        assert np.allclose(gw1[b].cpu().numpy(), ref_gw, rtol=1e-4, atol=1e-5)
        assert np.allclose(gg1[b].cpu().numpy(), ref_gg, rtol=1e-3, atol=2e-5)
    with pytest.raises(RuntimeError):
        denoiser.filtering_autograd(w, g, x, requires_grad=False).sum().backward()


@pytest.mark.gpu
def test_guidance_net_trains_through_the_filter():
    """the reference's training step shape (runner.py:70-80): model.filtering(aux, img, requires_grad=True),
    a loss, backward, one optimiser step -- the loss goes down and every parameter gets a gradient"""
    from rt_octree_amd import denoiser
    torch.manual_seed(0)
    model = denoiser.GuidanceNet(8, 16, 2, 2, 3).to("cuda:0")
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    rs = np.random.RandomState(0)
    clean = np.clip(rs.rand(1, 32, 48, 1) * 0.2 + np.linspace(0, 0.8, 48)[None, None, :, None], 0, 1).repeat(4, -1).astype(np.float32)
    clean[..., 3] = 1
    noisy = clean + rs.randn(*clean.shape).astype(np.float32) * 0.15
    noisy[..., 3] = 1
    aux = np.concatenate([noisy[..., :3], np.ones_like(noisy[..., :1]), noisy[..., :3] ** 2, np.ones_like(noisy[..., :1])], -1)
    aux_t = _dev(aux.transpose(0, 3, 1, 2))
    x, target = _dev(noisy), _dev(clean)
    losses = []
    for it in range(12):
        opt.zero_grad()
        out = denoiser.filtering(model, aux_t, x, requires_grad=True)
        loss = ((out[..., :3] - target[..., :3]) ** 2).mean()
        loss.backward()
        if it == 0:
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]


@pytest.mark.gpu
def test_hip_filter_matches_committed_golden():
    """tests/golden/kat_golden.npz f.*: the committed L = 4, 48x40 filter vectors (output, saved tensors,
    gradients) through the HIP kernels, bit for bit -- no oracle involved at test time."""
    import os
    from rt_octree_amd._lib import check, lib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_golden.npz"))
    L, H, W = g["f.guidance"].shape
    w, gd, x, go = _dev(g["f.weight"][None]), _dev(g["f.guidance"][None]), _dev(g["f.noisy"][None]), _dev(g["f.grad_out"][None])
    out = torch.empty_like(x)
    rf = torch.empty((1, L, H, W, 4), device="cuda:0")
    mx, inv = torch.empty((1, L, H, W), device="cuda:0"), torch.empty((1, L, H, W), device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    check(lib().rto_filtering_train_forward(s, w.data_ptr(), gd.data_ptr(), L, H, W, 1, x.data_ptr(), out.data_ptr(),
                                            rf.data_ptr(), mx.data_ptr(), inv.data_ptr()))
    gw, gg = torch.empty_like(w), torch.empty_like(gd)
    check(lib().rto_filtering_backward(s, go.data_ptr(), x.data_ptr(), w.data_ptr(), gd.data_ptr(), rf.data_ptr(),
                                       mx.data_ptr(), inv.data_ptr(), L, H, W, 1, gw.data_ptr(), gg.data_ptr()))
    out2 = torch.empty_like(x)
    check(lib().rto_filtering_batch(s, w.data_ptr(), gd.data_ptr(), L, H, W, 1, x.data_ptr(), out2.data_ptr()))
    torch.cuda.synchronize()
    for name, got in (("out", out), ("out", out2), ("rgb_filtered", rf), ("max_map", mx), ("inv_kernel_sum", inv),
                      ("grad_weight", gw), ("grad_guidance", gg)):
        assert np.array_equal(got[0].cpu().numpy().view(np.uint32), g["f." + name].view(np.uint32)), name
