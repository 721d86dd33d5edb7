"""Development harness: the fused GuidanceNet kernel alone (the bench's route: squares implied, packed fp16 maps) and
the factorised filter behind it, ms per 50-frame batch at 800x800."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rt_octree_amd import denoiser  # noqa: E402


def main():
    n, H, W = 50, 800, 800
    torch.manual_seed(0)
    full = denoiser.GuidanceNet(8, 32, 5, 2, 4)
    net = denoiser.FusedGuidanceNet(denoiser.GuidanceNetCompact.from_full(full).eval(), device=0)
    aux = torch.rand(n, 8, H, W, device="cuda:0")
    aux[:, 4:] = aux[:, :4] * aux[:, :4]
    img = torch.rand(n, H, W, 4, device="cuda:0")
    out = torch.empty_like(img)
    for rep in range(3):
        net(aux)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            w, g = net(aux)
        e1.record()
        torch.cuda.synchronize()
        print("guidance_fused (fp32 planes): %.3f ms per %d frames" % (e0.elapsed_time(e1) / 20, n), flush=True)
        e0.record()
        for _ in range(20):
            net.forward_packed(aux, squares_implied=True)
        e1.record()
        torch.cuda.synchronize()
        t_net = e0.elapsed_time(e1) / 20
        e0.record()
        for _ in range(20):
            net.filter_packed(img, out)
        e1.record()
        torch.cuda.synchronize()
        print("guidance_fused (packed, squares implied): %.3f ms   filter_fast (packed): %.3f ms per %d frames"
              % (t_net, e0.elapsed_time(e1) / 20, n), flush=True)
    print("checksum %.6f %.6f %.6f" % (float(w.double().sum()), float(g.double().sum()), float(out.double().sum())))


if __name__ == "__main__":
    main()
