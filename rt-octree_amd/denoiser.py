"""GuidanceNet denoise stage on PyTorch-ROCm + the HIP guided filter.

Reference (relative to /root/reference):
    denoiser/network.py:49-75    RepVGGBlock        (num_branches 3x3 + num_branches 1x1 [+ identity], ReLU6)
    denoiser/network.py:86-121   GuidanceNet        (layers; softmax over the first kernel_levels channels)
    denoiser/network.py:123-168  RepVGGBlockCompact / GuidanceNetCompact (branches folded into one 3x3)
    denoiser/network.py:170-208  compact_and_compile (fp16 + torch.jit.trace of `compact(aux.half())`)
    renderer/src/denoiser/denoiser.cpp:31-61  Denoiser::denoise (wrap aux, forward, filtering)

The network is small dense convolution work that north_star leaves to PyTorch-ROCm (MIOpen);
the filter that applies its output is the hand-written kernel in csrc/filter_kernels.hip.
Parameter names match the reference's modules, so its checkpoints (`state_dict`) and its
TorchScript exports (`ts_*.ts`) load unchanged.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import volrend as V


class RepVGGBlock(nn.Module):
    """network.py:49-75"""

    def __init__(self, in_channels, out_channels, num_branches):
        super().__init__()
        self.in_channels, self.out_channels, self.num_branches = in_channels, out_channels, num_branches
        self.conv3 = nn.ModuleList(nn.Conv2d(in_channels, out_channels, 3, padding="same") for _ in range(num_branches))
        self.conv1 = nn.ModuleList(nn.Conv2d(in_channels, out_channels, 1, padding="same") for _ in range(num_branches))

    def forward(self, x):
        # summation order of network.py:66-74: last 3x3 first, then the other 3x3s, the 1x1s, identity
        h = self.conv3[self.num_branches - 1](x)
        for i in range(self.num_branches - 1):
            h = h + self.conv3[i](x)
        for i in range(self.num_branches):
            h = h + self.conv1[i](x)
        if self.in_channels == self.out_channels:
            h = h + x
        return F.relu6(h)


class GuidanceNet(nn.Module):
    """network.py:86-121.  forward(aux [B,8,H,W]) -> (weight_map [B,L,H,W], guidance_map [B,L,H,W])"""

    def __init__(self, in_channels=8, mid_channels=32, num_branches=5, num_layers=2, kernel_levels=4):
        super().__init__()
        self.in_channels, self.mid_channels = in_channels, mid_channels
        self.num_branches, self.num_layers, self.kernel_levels = num_branches, num_layers, kernel_levels
        self.layers = nn.ModuleList(self._make_layers(RepVGGBlock))

    def _make_layers(self, block, *extra):
        layers = []
        for i in range(self.num_layers - 1):
            layers.append(block(self.mid_channels if i > 0 else self.in_channels, self.mid_channels, self.num_branches, *extra))
        layers.append(block(self.mid_channels if self.num_layers > 1 else self.in_channels,
                            self.kernel_levels * 2, self.num_branches, *extra))
        return layers

    def features(self, x):
        for layer in self.layers:
            x = layer(x)
        return x

    def split(self, x):
        """network.py:111-118"""
        x = x.float()
        weight_map = F.softmax(x[:, :self.kernel_levels, ...].contiguous(), dim=1)
        guidance_map = x[:, self.kernel_levels:, ...].contiguous()
        return weight_map, guidance_map

    def forward(self, aux_buffer):
        if aux_buffer.is_cuda:
            with torch.autocast("cuda", dtype=torch.float16):  # network.py:104-108
                x = self.features(aux_buffer)
        else:
            x = self.features(aux_buffer)
        return self.split(x)


class RepVGGBlockCompact(nn.Module):
    """network.py:123-154: one 3x3 conv = sum of the 3x3 branches + zero-padded 1x1 branches
    (+ identity when Cin == Cout)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 3, padding="same")

    @classmethod
    def from_full(cls, full):
        blk = cls(full.in_channels, full.out_channels)
        with torch.no_grad():
            w = torch.zeros_like(blk.conv.weight)
            b = torch.zeros_like(blk.conv.bias)
            for i in range(full.num_branches):
                w += full.conv3[i].weight
                b += full.conv3[i].bias
            for i in range(full.num_branches):
                w += F.pad(full.conv1[i].weight, (1, 1, 1, 1))
                b += full.conv1[i].bias
            if full.in_channels == full.out_channels:
                for i in range(full.out_channels):
                    w[i, i % full.in_channels, 1, 1] += 1
            blk.conv.weight.copy_(w)
            blk.conv.bias.copy_(b)
        blk.conv.requires_grad_(False)
        return blk

    def forward(self, x):
        return F.relu6(self.conv(x))


class GuidanceNetCompact(GuidanceNet):
    """network.py:156-168: the inference network (plain conv3x3 + ReLU6 stack)."""

    def __init__(self, in_channels=8, mid_channels=32, num_layers=2, kernel_levels=4):
        nn.Module.__init__(self)
        self.in_channels, self.mid_channels = in_channels, mid_channels
        self.num_branches, self.num_layers, self.kernel_levels = 1, num_layers, kernel_levels
        chans = [in_channels] + [mid_channels] * (num_layers - 1) + [kernel_levels * 2]
        self.layers = nn.ModuleList(RepVGGBlockCompact(chans[i], chans[i + 1]) for i in range(num_layers))

    @classmethod
    def from_full(cls, full):
        net = cls(full.in_channels, full.mid_channels, full.num_layers, full.kernel_levels)
        net.layers = nn.ModuleList(RepVGGBlockCompact.from_full(l) for l in full.layers)
        return net.eval()

    def forward(self, aux_buffer):
        # compact_and_compile's traced function: compact.forward(aux.half()) with fp16 weights
        dtype = self.layers[0].conv.weight.dtype
        return self.split(self.features(aux_buffer.to(dtype)))


def compact_and_compile(model, device=None, example_hw=(800, 800)):
    """network.py:170-208: fold, cast to fp16 (on GPU), trace to TorchScript.  On CPU the fold is
    kept in fp32 (fp16 convolution is a GPU path)."""
    compact = GuidanceNetCompact.from_full(model.eval().cpu())
    dev = torch.device(device) if device is not None else torch.device("cpu")
    compact = compact.to(dev)
    if dev.type == "cuda":
        compact = compact.half()
    aux = torch.rand((1, model.in_channels) + tuple(example_hw), device=dev)
    with torch.no_grad():
        return torch.jit.trace(compact, (aux,), check_trace=False)


class FusedGuidanceNet:
    """The compact GuidanceNet as ONE hand-written gfx950 kernel (csrc/guidance_kernels.hip, MFMA fp16
    with fp32 accumulation) instead of two MIOpen convolutions plus seven elementwise launches.
    Built from a GuidanceNetCompact, from a TorchScript trace of that module (parameters named
    layers.<i>.conv.weight / .bias), or from a ts_*.ts as the REFERENCE's exporter writes it -- a traced
    closure without parameters whose conv weights are constants of the graph (network.py:194-201);
    callable like the module: aux [n,8,H,W] float32 cuda -> (weight_map, guidance_map) [n,L,H,W] float32."""

    @staticmethod
    def graph_conv_constants(ts_module):
        """{weight, bias} tensors of the convolution nodes of a TorchScript forward graph, in order."""
        g = ts_module.forward.inlined_graph if hasattr(ts_module.forward, "inlined_graph") else ts_module.inlined_graph
        out = []
        for n in g.nodes():
            if n.kind() in ("aten::_convolution_mode", "aten::conv2d", "aten::_convolution", "aten::convolution"):
                ins = list(n.inputs())
                w, b = ins[1].toIValue(), ins[2].toIValue()
                if not (torch.is_tensor(w) and torch.is_tensor(b)):
                    return []
                out.append((w, b))
        return out

    def __init__(self, module, device=0):
        import ctypes as C
        from ._lib import check, lib
        sd = {k: v.detach().float().cpu().contiguous() for k, v in module.state_dict().items()}
        keys = ("layers.0.conv.weight", "layers.0.conv.bias", "layers.1.conv.weight", "layers.1.conv.bias")
        if not sd and isinstance(module, torch.jit.ScriptModule):  # reference-format export
            convs = self.graph_conv_constants(module)
            if len(convs) == 2:
                sd = {k: v.detach().float().cpu().contiguous()
                      for k, v in zip(keys, (convs[0][0], convs[0][1], convs[1][0], convs[1][1]))}
        if not all(k in sd for k in keys) or any(k.startswith("layers.2.") for k in sd):
            raise ValueError("FusedGuidanceNet needs a two-layer compact GuidanceNet (layers.{0,1}.conv.*)")
        w1, b1, w2, b2 = (sd[k] for k in keys)
        if tuple(w1.shape[1:]) != (8, 3, 3) or tuple(w2.shape[2:]) != (3, 3) or w2.shape[1] != w1.shape[0]:
            raise ValueError("unexpected GuidanceNet weight shapes")
        self.c1, self.levels = int(w1.shape[0]), int(w2.shape[0]) // 2
        self.device = torch.device("cuda", device)
        h = C.c_void_p(0)
        check(lib().rto_guidance_net_create(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), self.c1,
                                            self.levels, device, C.byref(h)))
        self._h = h
        self._out = {}

    def __call__(self, aux, stream=None, squares_implied=False, cull=None, rgba=False):
        """squares_implied: aux planes 4..7 are the fp32 squares of planes 0..3 (the renderer's aux buffer): the
        kernel reads half the bytes, results are bit-identical.  cull = RenderContext.tile_marks() of the launch that
        rendered aux: tiles whose inputs are all background get the background maps without being computed (same bits).
        rgba: `aux` is the noisy image [n, H, W, 4] = (r, g, b, alpha) of a lean batched launch (RTO_NET_INPUT_RGBA)"""
        from ._lib import check, lib
        if rgba:
            n, H, W, c = aux.shape
            assert c == 4 and aux.dtype == torch.float32 and aux.is_contiguous()
        else:
            n, c, H, W = aux.shape
            assert c == 8 and aux.dtype == torch.float32 and aux.is_contiguous()
        key = (n, H, W)
        if key not in self._out:
            self._out[key] = (torch.empty((n, self.levels, H, W), device=self.device),
                              torch.empty((n, self.levels, H, W), device=self.device))
        wm, gm = self._out[key]
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        marks, words, bg = (None, 0, 0.0)
        if cull is not None:
            marks, words, _slot0, frames, bg = cull
            if frames < n:
                raise ValueError("GuidanceNet: %d frames but tile marks of %d" % (n, frames))
        check(lib().rto_guidance_net_forward_culled(self._h, V._stream_ptr(s), aux.data_ptr(), n, H, W, wm.data_ptr(), gm.data_ptr(),
                                                    2 if rgba else (1 if squares_implied else 0), marks, int(words), float(bg)))
        return wm, gm

    def denoise(self, ctx, n=1, mode=V.FILTER_FAST, stream=None):
        """Denoiser::denoise (denoiser.cpp:31-61) in one call for the n frames of `ctx` from its selected slot on
        (rto_denoise): network on ctx.aux, filter ctx.noisy -> ctx.image; uses the tile marks of the batched launch that
        rendered them when there was one"""
        from ._lib import check, lib
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        check(lib().rto_denoise(self._h, ctx._h, int(n), int(mode), V._stream_ptr(s)))

    def filter_planes(self, weight_map, guidance_map, img_in, img_out, mode=V.FILTER_EXACT, stream=None, cull=None):
        """volrend.filtering on this network's fp32 maps [n, L, H, W], skipping the filter tiles that see only culled render
        tiles (cull = RenderContext.tile_marks(); rto_filtering_culled; same bits).  cull=None: volrend.filtering"""
        from ._lib import check, lib
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        n, L, H, W = (int(x) for x in guidance_map.shape)
        marks, words, bg = (None, 0, 0.0)
        if cull is not None:
            marks, words, _slot0, frames, bg = cull
            if frames < n:
                raise ValueError("filter_planes: %d images but tile marks of %d frames" % (n, frames))
        check(lib().rto_filtering_culled(self._h, V._stream_ptr(s), V._dev_ptr(weight_map), V._dev_ptr(guidance_map), H, W, n,
                                         V._dev_ptr(img_in), V._dev_ptr(img_out), int(mode), marks, int(words), float(bg)))

    def reserve(self, n, H, W):
        """size the packed-map scratch up front (growing it later synchronises the device once)"""
        from ._lib import check, lib
        check(lib().rto_guidance_net_reserve(self._h, int(n), int(H), int(W)))

    def forward_packed(self, aux, stream=None, squares_implied=False, cull=None, rgba=False, sparse=False):
        """the network, its 8 fp16 output channels kept packed in the handle's scratch (rto_guidance_net_forward_packed).
        cull = RenderContext.tile_marks() of the launch that rendered aux (frames in order): network tiles whose inputs are
        all background get the network's background output without being computed (same bits).
        rgba: `aux` is the noisy image [n, H, W, 4] = (r, g, b, alpha) of a lean batched launch (RTO_NET_INPUT_RGBA)
        sparse (with rgba and cull): ... of a SPARSE lean launch (RenderContext.set_lean_outputs(2)): pixels of unmarked tiles are
        taken as background, the skipped tiles' maps are not stored; filter_packed with the same marks completes the route"""
        from ._lib import check, lib
        if rgba:
            n, H, W, c = aux.shape
            assert c == 4 and aux.dtype == torch.float32 and aux.is_contiguous()
        else:
            n, c, H, W = aux.shape
            assert c == 8 and aux.dtype == torch.float32 and aux.is_contiguous()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        marks, words, bg = (None, 0, 0.0)
        if cull is not None:
            marks, words, _slot0, frames, bg = cull
            if frames < n:
                raise ValueError("forward_packed: %d frames but tile marks of %d" % (n, frames))
        check(lib().rto_guidance_net_forward_packed_culled(self._h, V._stream_ptr(s), aux.data_ptr(), n, H, W,
                                                           (2 if rgba else (1 if squares_implied else 0)) | (4 if sparse else 0),
                                                           marks, int(words), float(bg)))
        self._packed_shape = (n, H, W)

    def filter_packed(self, img_in, img_out, stream=None, shape=None, cull=None):
        """factorised filter on the packed maps of the last forward_packed: img_in / img_out device pointers or tensors
        of `shape` = (n, H, W) images (default: the extent of that forward; the library refuses any other).
        cull = RenderContext.tile_marks() of the launch that rendered img_in (images = the marks' frames, in order): tiles
        that see only background are copied, not filtered (rto_filtering_packed_culled; same bits)"""
        from ._lib import check, lib
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        n, H, W = shape if shape is not None else self._packed_shape
        if cull is None:
            check(lib().rto_filtering_packed(self._h, V._stream_ptr(s), V._dev_ptr(img_in), V._dev_ptr(img_out), int(n), int(H), int(W)))
            return
        marks, words, _slot0, frames, bg = cull
        if frames < n:
            raise ValueError("filter_packed: %d images but tile marks of %d frames" % (n, frames))
        check(lib().rto_filtering_packed_culled(self._h, V._stream_ptr(s), V._dev_ptr(img_in), V._dev_ptr(img_out), int(n), int(H), int(W),
                                                marks, int(words), float(bg)))

    def __del__(self):
        try:
            from ._lib import lib
            if getattr(self, "_h", None):
                lib().rto_guidance_net_free(self._h)
        except Exception:
            pass


class _Filtering(torch.autograd.Function):
    """denoiser::Filtering (filtering.cu:596-707): the guided filter under autograd.  Forward runs the
    fused HIP kernel (saving rgb_filtered / max_map / inv_kernel_sum per level when a gradient is
    wanted), backward the gather-form HIP kernel; both through the C ABI on torch's current stream."""

    @staticmethod
    def forward(ctx, weight_map, guidance_map, img_in, requires_grad):
        from ._lib import check, lib
        if not (weight_map.is_cuda and guidance_map.is_cuda and img_in.is_cuda):
            raise RuntimeError("filtering_autograd needs CUDA(HIP) tensors: the filter has no CPU path")
        w = weight_map.detach().float().contiguous()
        g = guidance_map.detach().float().contiguous()
        x = img_in.detach().float().contiguous()
        B, L, H, W = g.shape
        if w.shape != g.shape or tuple(x.shape) != (B, H, W, 4):
            raise RuntimeError("filtering_autograd: weight/guidance [B,L,H,W] and img_in [B,H,W,4] expected")
        out = torch.empty_like(x)
        s = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            if requires_grad:
                rf = torch.empty((B, L, H, W, 4), device=x.device, dtype=torch.float32)
                mx = torch.empty((B, L, H, W), device=x.device, dtype=torch.float32)
                inv = torch.empty((B, L, H, W), device=x.device, dtype=torch.float32)
                check(lib().rto_filtering_train_forward(s, w.data_ptr(), g.data_ptr(), L, H, W, B, x.data_ptr(),
                                                        out.data_ptr(), rf.data_ptr(), mx.data_ptr(), inv.data_ptr()))
                ctx.save_for_backward(w, g, x, rf, mx, inv)
            else:
                check(lib().rto_filtering_batch(s, w.data_ptr(), g.data_ptr(), L, H, W, B, x.data_ptr(), out.data_ptr()))
        ctx.has_saved = bool(requires_grad)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        from ._lib import check, lib
        if not ctx.has_saved:
            raise RuntimeError("filtering_autograd was called with requires_grad=False")
        w, g, x, rf, mx, inv = ctx.saved_tensors
        go = grad_output.float().contiguous()
        B, L, H, W = g.shape
        gw, gg = torch.empty_like(w), torch.empty_like(g)
        s = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            check(lib().rto_filtering_backward(s, go.data_ptr(), x.data_ptr(), w.data_ptr(), g.data_ptr(), rf.data_ptr(),
                                               mx.data_ptr(), inv.data_ptr(), L, H, W, B, gw.data_ptr(), gg.data_ptr()))
        return gw, gg, None, None  # (:701-706: no gradient for img_in / the flag)


def filtering_autograd(weight_map, guidance_map, imgs_in, requires_grad=False):
    """`_denoiser.filtering_autograd` (denoiser/extension/bindings.cpp): weight_map, guidance_map
    [B,L,H,W], imgs_in [B,H,W,4] -> filtered images [B,H,W,4]."""
    return _Filtering.apply(weight_map, guidance_map, imgs_in, requires_grad)


def filtering(model, aux_buffer, img_in, requires_grad=False):
    """denoiser/network.py:77-84: model(aux) -> (weight_map, guidance_map) -> filtering_autograd."""
    weight_map, guidance_map = model(aux_buffer)
    return filtering_autograd(weight_map, guidance_map, img_in, requires_grad=requires_grad)


class Denoiser:
    """volrend::Denoiser (denoiser.hpp:11-21, denoiser.cpp:31-61).

    `Denoiser(path)` loads a TorchScript module exactly like the reference; a torch.nn.Module with
    the (weight_map, guidance_map) contract is accepted too.  An empty path raises the reference's
    "No torchscript module is given to denoiser." (denoiser.cpp:13-16)."""

    def __init__(self, ts_module, device=0):
        self.device = torch.device("cuda", device)
        if isinstance(ts_module, (str, bytes)):
            if not ts_module:
                raise RuntimeError("No torchscript module is given to denoiser.")
            try:
                self.module = torch.jit.load(ts_module, map_location=self.device)
            except Exception as e:  # denoiser.cpp:22-26
                raise RuntimeError("Error when loading torchscript model from %s" % ts_module) from e
        else:
            self.module = ts_module.to(self.device)
        self.module.eval()

    @torch.no_grad()
    def denoise(self, cam, ctx, stream=None):
        """aux [1,8,H,W] (zero-copy) -> module -> filtering(noisy -> image).  Timer buckets
        torch / filter bracket the two stages like denoiser.cpp:36-60."""
        tm = ctx.timer()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(s) if hasattr(s, "cuda_stream") else _null():
            tm.torch_start()
            aux = torch.as_tensor(ctx.aux_view(), device=self.device)
            weight_map, guidance_map = self.module(aux)
            weight_map = weight_map.squeeze(0).contiguous()
            guidance_map = guidance_map.squeeze(0).contiguous()
            tm.torch_stop()
            tm.filter_start()
            V.filtering(s, weight_map, guidance_map, ctx.noisy_ptr, ctx.image_ptr)
            tm.filter_stop()
        return weight_map, guidance_map


class _null:
    def __enter__(self): return self
    def __exit__(self, *a): return False
