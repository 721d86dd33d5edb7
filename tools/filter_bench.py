"""Development harness: time the fused guided filter (rto_filtering_batch) on random maps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rt_octree_amd as R  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for n, H, W, L in ((8, 800, 800, 4), (1, 800, 800, 4), (2, 1080, 1920, 4)):
        g = torch.randn(n, L, H, W, device=dev) * 3
        w = torch.softmax(torch.randn(n, L, H, W, device=dev), 1).contiguous()
        img = torch.rand(n, H, W, 4, device=dev)
        out = torch.empty_like(img)
        s = torch.cuda.current_stream()
        for _ in range(3):
            R.filtering(s, w, g, img, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        reps = 20
        for _ in range(reps):
            R.filtering(s, w, g, img, out)
        e1.record(s)
        torch.cuda.synchronize()
        print("filter n=%d %dx%d L=%d: %.3f ms/image" % (n, W, H, L, e0.elapsed_time(e1) / reps / n), flush=True)
        g6 = g.clamp(0, 6).contiguous()
        for _ in range(3):
            R.filtering(s, w, g6, img, out, mode=R.FILTER_FAST)
        torch.cuda.synchronize()
        e0.record(s)
        for _ in range(reps):
            R.filtering(s, w, g6, img, out, mode=R.FILTER_FAST)
        e1.record(s)
        torch.cuda.synchronize()
        print("  factorised: %.3f ms/image" % (e0.elapsed_time(e1) / reps / n), flush=True)
        # training side: forward with saves + backward through the autograd wrapper's ABI calls
        from rt_octree_amd import denoiser
        wr, gr = w.clone().requires_grad_(True), g.clone().requires_grad_(True)
        go = torch.randn_like(img)
        for timed in (False, True):
            if timed:
                torch.cuda.synchronize()
                e0.record(s)
            for _ in range(5):
                o = denoiser.filtering_autograd(wr, gr, img, requires_grad=True)
                o.backward(go)
                wr.grad = gr.grad = None
            if timed:
                e1.record(s)
                torch.cuda.synchronize()
                print("  train forward + backward: %.3f ms/image" % (e0.elapsed_time(e1) / 5 / n), flush=True)


if __name__ == "__main__":
    main()
