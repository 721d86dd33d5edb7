"""GuidanceNet / GuidanceNetCompact (PyTorch) against golden tensors produced by the imported
reference module denoiser/network.py (tests/golden/make_goldens.py).  CPU, fp32.
Tolerances: full net 1e-6 (same ops, same order); folded net 5e-6 (re-associated conv sums)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from rt_octree_amd import denoiser  # noqa: E402

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "guidance_golden.npz"))


def _full():
    net = denoiser.GuidanceNet(8, 32, 5, 2, 4)
    sd = {k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("sd.")}
    net.load_state_dict(sd, strict=True)  # parameter names are the reference's
    return net.eval()


def test_full_network_matches_reference():
    net = _full()
    assert sum(p.numel() for p in net.parameters()) == 26000  # SURVEY 8c probe
    with torch.no_grad():
        w, g = net(torch.from_numpy(G["aux"]))
    assert np.allclose(w.numpy(), G["weight_full"], atol=1e-6) and np.allclose(g.numpy(), G["guidance_full"], atol=1e-6)
    assert np.allclose(w.sum(1).numpy(), 1, atol=1e-6)  # softmax over the kernel levels


def test_compact_fold_matches_reference():
    net = _full()
    c = denoiser.GuidanceNetCompact.from_full(net)
    assert sum(p.numel() for p in c.parameters()) == 4648
    for k in G.files:
        if k.startswith("csd."):
            assert np.allclose(c.state_dict()[k[4:]].numpy(), G[k], atol=1e-6), k
    with torch.no_grad():
        w, g = c(torch.from_numpy(G["aux"]))
    assert np.allclose(w.numpy(), G["weight_compact"], atol=5e-6) and np.allclose(g.numpy(), G["guidance_compact"], atol=5e-6)
    assert np.allclose(w.numpy(), G["weight_full"], atol=5e-6)
    # the reference's compact state_dict loads into ours by name
    c2 = denoiser.GuidanceNetCompact(8, 32, 2, 4)
    c2.load_state_dict({k[4:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("csd.")}, strict=True)


def test_torchscript_export_roundtrip(tmp_path):
    ts = denoiser.compact_and_compile(_full(), device=None, example_hw=(24, 20))
    p = str(tmp_path / "ts_latest.ts")
    ts.save(p)
    m = torch.jit.load(p)
    with torch.no_grad():
        w, g = m(torch.from_numpy(G["aux"]))
    assert np.allclose(w.numpy(), G["weight_compact"], atol=5e-6) and np.allclose(g.numpy(), G["guidance_compact"], atol=5e-6)


def test_denoiser_requires_module():
    with pytest.raises(RuntimeError) as e:
        denoiser.Denoiser("")
    assert "No torchscript module is given to denoiser." in str(e.value)


def test_committed_trained_weights_load_into_the_reference_layout():
    """rt-octree_amd/weights/guidance_synth_lego.pt (tools/train_guidance.py): a state_dict with the
    reference module's parameter names for GuidanceNet(8, 32, 5, 2, 4); folds into the compact net."""
    import os
    import torch
    from rt_octree_amd import denoiser
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rt-octree_amd", "weights",
                        "guidance_synth_lego.pt")
    sd = torch.load(path, map_location="cpu")
    full = denoiser.GuidanceNet(8, 32, 5, 2, 4)
    full.load_state_dict(sd)  # strict
    assert all(torch.isfinite(v).all() for v in sd.values())
    assert "layers.0.conv3.4.weight" in sd and "layers.1.conv1.0.bias" in sd
    compact = denoiser.GuidanceNetCompact.from_full(full).eval()
    aux = torch.rand(1, 8, 24, 32)
    with torch.no_grad():
        w0, g0 = full.eval()(aux)
        w1, g1 = compact(aux)
    assert torch.allclose(w0, w1, atol=1e-4) and torch.allclose(g0, g1, atol=1e-3)


def test_reference_format_ts_fixture_is_recognised():
    """tests/golden/ts_ref_format.ts was written by the reference's OWN exporter (compact_and_compile traces a
    closure, network.py:194-201; torch.jit.save, runner.py:171-175): a module without parameters whose
    conv weights are graph constants.  The host finds them and they equal this repository's fold of the
    same checkpoint bit for bit."""
    import os
    import torch
    from rt_octree_amd import denoiser
    here = os.path.dirname(os.path.abspath(__file__))
    m = torch.jit.load(os.path.join(here, "golden", "ts_ref_format.ts"), map_location="cpu")
    assert list(m.named_parameters()) == [] and len(m.state_dict()) == 0
    convs = denoiser.FusedGuidanceNet.graph_conv_constants(m)
    assert [tuple(w.shape) for w, _ in convs] == [(32, 8, 3, 3), (8, 32, 3, 3)]
    full = denoiser.GuidanceNet(8, 32, 5, 2, 4)
    full.load_state_dict(torch.load(os.path.join(here, "..", "rt-octree_amd", "weights", "guidance_synth_lego.pt"), map_location="cpu"))
    sd = denoiser.GuidanceNetCompact.from_full(full).half().state_dict()
    for i, (w, b) in enumerate(convs):
        assert torch.equal(w, sd["layers.%d.conv.weight" % i]) and torch.equal(b, sd["layers.%d.conv.bias" % i])
    # and the traced graph computes what the compact network computes (fp16 convolutions on the CPU)
    torch.manual_seed(1)
    aux = torch.rand(1, 8, 20, 24)
    with torch.no_grad():
        w_ts, g_ts = m(aux)
        w_c, g_c = denoiser.GuidanceNetCompact.from_full(full).half()(aux.half())
    assert torch.allclose(w_ts, w_c.float(), atol=2e-3) and torch.allclose(g_ts, g_c.float(), atol=1e-2)
