"""Run-to-run determinism of every kernel of the path while OTHER processes keep the same GPU busy (the GPU time-slices
the processes' queues: waves are saved and restored mid-kernel).  Each stage is run twice on the same inputs and compared
bit for bit, over random frame sizes; tools/contention_check.sh starts 8 copies of this script at once.

Why this exists (round 3): with 8 processes on one GPU, rto_filtering (the bit-exact filter, filter_fused) returned
different bits in ~25 % of its runs -- lanes 48..63 of some waves a few ulp to 1e-2 off.  Alone on the GPU everything was
deterministic, and so were all other kernels.  tools/scratch_hazard_probe.py isolated it: the build of that kernel on v_pk_fma_f32
(no other kernel had the instruction) goes wrong while waves of another process run MFMA-dense kernels.  The filter is built from scalar FMAs now
(tests/test_codegen.py keeps the instruction out) and this check stays.

python tools/contention_determinism.py SEED [ITERS]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import denoiser, synth  # noqa: E402


def main():
    import torch
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    basis = int(os.environ.get("RTO_CD_BASIS", "9"))  # 25: the shading kernel's SH25 instantiation spills (scratch) as well
    t = synth.make_tree(depth_limit=7, basis_dim=basis, shell=2.5)
    dt = R.N3Tree.from_arrays(t.child, t.data, t.scale, t.offset, t.data_format)
    torch.manual_seed(3)
    net = denoiser.FusedGuidanceNet(denoiser.GuidanceNetCompact.from_full(denoiser.GuidanceNet(8, 32, 5, 2, 4)).eval())
    rs = np.random.RandomState(seed)
    stages = ("render", "net planes", "net planes (all 8 aux planes read)", "net packed + factorised filter",
              "net packed (all 8 aux planes read) + factorised filter", "exact filter", "factorised filter on planes", "one-call denoise exact")
    cnt = dict.fromkeys(stages, 0)
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
        opt = R.RenderOptions(spp=6, denoise=True, background_brightness=float(rs.choice([1.0, 0.0, rs.rand()])))
        ctx = R.RenderContext(W, H, frames=n)
        aux = torch.as_tensor(ctx.batch_views()[0], device="cuda:0")
        img = torch.as_tensor(ctx.batch_views()[2], device="cuda:0")
        ctx.select_frame(0)

        def twice(fn):
            outs = []
            for _ in range(2):
                fn()
                torch.cuda.synchronize()
                outs.append(img[:n].clone())
            return torch.equal(*outs)

        def render():
            ctx.rng_seed()
            R.launch_renderer_batch(dt, cams, opt, ctx)
            img[:n].copy_(aux[:n, :4].permute(0, 2, 3, 1))

        def planes(sq):
            def f():
                w, g = net(aux[:n], squares_implied=sq)
                img[:n].copy_((w + g).permute(0, 2, 3, 1))
            return f

        def packed(sq):
            def f():
                net.forward_packed(aux[:n], squares_implied=sq)
                net.filter_packed(ctx.noisy_ptr, ctx.image_ptr, shape=(n, H, W))
            return f

        cnt["render"] += not twice(render)
        cnt["net planes"] += not twice(planes(True))
        cnt["net planes (all 8 aux planes read)"] += not twice(planes(False))
        cnt["net packed + factorised filter"] += not twice(packed(True))
        cnt["net packed (all 8 aux planes read) + factorised filter"] += not twice(packed(False))
        w1, g1 = (x.clone() for x in net(aux[:n], squares_implied=True))
        cnt["exact filter"] += not twice(lambda: R.filtering(None, w1, g1, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_EXACT))
        cnt["factorised filter on planes"] += not twice(lambda: R.filtering(None, w1, g1, ctx.noisy_ptr, ctx.image_ptr, mode=R.FILTER_FAST))
        cnt["one-call denoise exact"] += not twice(lambda: net.denoise(ctx, n, R.FILTER_EXACT))
        ctx.free()
    print("basis %d" % basis, end=" "); print("seed %d: runs that differed from their repeat, of %d: %s" % (seed, iters, cnt), flush=True)
    return 1 if any(cnt.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
