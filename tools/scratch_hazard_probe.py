"""Which property of an earlier kernel makes filter_fused non-deterministic when processes share the GPU?  (See
tools/contention_determinism.py for the finding.)  Per iteration: a fresh frame, one launch of rto_probe_scratch(kind)
-- kind bit 0: a private segment (208 B per lane of scratch), bit 1: 34 KB of static LDS, bit 2: an MFMA loop -- then the
bit-exact filter twice on the same inputs, compared.  Start 8 copies at once (tools/scratch_hazard_probe.sh).
python tools/scratch_hazard_probe.py SEED KIND [ITERS]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import denoiser, synth  # noqa: E402
from rt_octree_amd._lib import check, lib  # noqa: E402


def main():
    import torch
    seed, kind = int(sys.argv[1]), int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    t = synth.make_tree(depth_limit=7, basis_dim=9, shell=2.5)
    dt = R.N3Tree.from_arrays(t.child, t.data, t.scale, t.offset, t.data_format)
    torch.manual_seed(3)
    net = denoiser.FusedGuidanceNet(denoiser.GuidanceNetCompact.from_full(denoiser.GuidanceNet(8, 32, 5, 2, 4)).eval())
    rs = np.random.RandomState(seed)
    bad = 0
    for _ in range(iters):
        W, H = int(rs.randint(40, 520)), int(rs.randint(40, 420))
        n = int(rs.randint(1, 5))
        fx = float(rs.uniform(0.6, 2.5) * W)
        cams = []
        for _ in range(n):
            pos = rs.randn(3)
            pos = pos / np.linalg.norm(pos) * rs.uniform(2.5, 5.0)
            c = R.Camera(W, H, fx, fx)
            c.set_c2w(synth.look_at_c2w(pos, rs.uniform(-0.4, 0.4, 3)))
            cams.append(c)
        ctx = R.RenderContext(W, H, frames=n)
        ctx.rng_seed()
        R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=True), ctx)
        torch.cuda.synchronize()
        aux = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
        img = torch.as_tensor(ctx.batch_views()[2], device="cuda:0")
        ctx.select_frame(0)
        w1, g1 = (x.clone() for x in net(aux[:n], squares_implied=True))
        torch.cuda.synchronize()
        if kind >= 0:
            check(lib().rto_probe_scratch(kind, 4096, 64))
            torch.cuda.synchronize()
        victim = os.environ.get("RTO_SHP_VICTIM", "exact")

        def run_victim():
            if victim == "exact":
                R.filtering(None, w1, g1, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_EXACT)
            elif victim == "fast_planes":
                R.filtering(None, w1, g1, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_FAST)
            elif victim == "packed":
                net.forward_packed(aux[:n], squares_implied=True)
                net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
            elif victim == "render":
                ctx.rng_seed()
                R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=False), ctx)
            elif victim == "net_planes":
                w, g = net(aux[:n], squares_implied=True)
                img[:n].copy_((w + g).permute(0, 2, 3, 1))

        outs = []
        for _ in range(2):
            run_victim()
            torch.cuda.synchronize()
            outs.append(img[:n].clone())
        bad += not torch.equal(*outs)
        ctx.free()
    what = "no probe" if kind < 0 else "+".join(w for b, w in ((1, "scratch"), (2, "LDS 34 KB"), (4, "MFMA")) if kind & b) or "plain kernel"
    print("probe kind %d (%s) victim %s seed %d: differed from its repeat in %d of %d iterations" % (kind, what, os.environ.get("RTO_SHP_VICTIM", "exact"), seed, bad, iters), flush=True)


if __name__ == "__main__":
    main()
