"""Trains a GuidanceNet on MI355X end to end with this repository's own pieces: noisy
SPP-6 buffers and 1024-spp targets rendered by the HIP renderer, the reference's training step
(denoiser/runner.py:70-84: model.filtering(aux, img, requires_grad=True) -> SMAPE loss -> backward ->
Adam, lr decayed 0.1^(it/N), AMP inside the model) through `denoiser.filtering_autograd`, i.e. the
HIP forward-with-saves and gather-form backward of the guided filter.

The result is a state_dict of the reference's module layout (layers.N.conv3/conv1.M.weight/bias), small
enough to commit: bench.py loads rt-octree_amd/weights/guidance_synth_lego.pt when it exists, so its PSNR
block reports a trained denoiser instead of random weights.  Held out of training: every pose with
index % 4 == 0 (pose 0 is the one bench.py scores).

usage: python tools/train_guidance.py [--iters 1500] [--out gpurun_out/guidance_synth_lego.pt]"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rt_octree_amd as R  # noqa: E402
from rt_octree_amd import denoiser, synth  # noqa: E402

WARM = 100  # frame i of the reference run renders with the RNG advanced (100 + i) times


def smape(pred, truth):  # denoiser/metrics.py:7-9
    return ((pred - truth).abs() / (pred.abs() + truth.abs() + 1e-5)).mean()


def psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--loss", default="smape", choices=["smape", "mse"])
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--basis", type=int, default=16)
    ap.add_argument("--shell", type=float, default=2.5)
    ap.add_argument("--target-frames", type=int, default=32, help="SPP-32 frames averaged into a target")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "guidance_synth_lego.pt"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    W = H = args.size

    key = "d%d_s%g_b%d" % (args.depth, args.shell, args.basis)
    tag = hashlib.sha1(open(os.path.join(ROOT, "rt-octree_amd", "synth.py"), "rb").read()).hexdigest()[:10]
    path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "rto_bench_tree_%s_%s.npz" % (key, tag))
    if not os.path.exists(path):  # same cache file as bench.py
        synth.make_tree(depth_limit=args.depth, basis_dim=args.basis, shell=args.shell).save_npz(path)
    tree = R.N3Tree(path)
    poses = synth.orbit_poses(200)
    fx = synth.blender_focal(W)

    def cam_of(i):
        c = R.Camera(W, H, fx, fx)
        c.set_c2w(poses[i])
        return c

    B = 16
    ctx = R.RenderContext(W, H, frames=B)
    stream = torch.cuda.current_stream(dev)
    aux_v, noisy_v, image_v = ctx.batch_views()
    aux_t, img_t = torch.as_tensor(aux_v, device=dev), torch.as_tensor(image_v, device=dev)

    def render(idx, spp, jumps):
        ctx.rng_seed()
        R.launch_renderer_batch(tree, [cam_of(i) for i in idx], R.RenderOptions(spp=spp, denoise=False), ctx, stream,
                                rng_jumps=jumps)
        torch.cuda.synchronize()
        return aux_t[:len(idx)].clone(), img_t[:len(idx)].clone()

    t0 = time.time()
    train_idx = [i for i in range(200) if i % 4 != 0][::3]   # 50 poses
    test_idx = [0, 40, 100, 160]
    data = {}
    for i in train_idx + test_idx:
        tgt = torch.zeros(H, W, 4, device=dev)
        for k in range(0, args.target_frames, B):
            n = min(B, args.target_frames - k)
            _, im = render([i] * n, 32, [5000 + 200 * i + k + j for j in range(n)])
            tgt += im.sum(0)
        tgt /= args.target_frames
        aux, noisy = render([i, i], 6, [WARM + i, 900 + i])  # the bench's realisation + a second one
        data[i] = (aux, noisy, tgt)
    print("dataset: %d train + %d test poses in %.1fs" % (len(train_idx), len(test_idx), time.time() - t0), flush=True)

    torch.manual_seed(0)
    model = denoiser.GuidanceNet(8, 32, 5, 2, 4).to(dev)  # denoiser/configs/blender.txt:21-25
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(0.9, 0.999), weight_decay=5e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / (args.iters + 1), 1))
    rs = np.random.RandomState(0)
    # the model runs its convolutions under fp16 autocast (network.py:104-108): without loss scaling the
    # per-pixel gradients of a mean over 2 M values underflow in the fp16 backward (runner.py:79 scales too)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 16)

    def evaluate(tag):
        model.eval()
        with torch.no_grad():
            raw, den = [], []
            for i in test_idx:
                aux, noisy, tgt = data[i]
                out = denoiser.filtering(model, aux[:1], noisy[:1])
                raw.append(psnr(noisy[0, ..., :3], tgt[..., :3]))
                den.append(psnr(out[0, ..., :3], tgt[..., :3]))
        model.train()
        print("%s: held-out PSNR raw %.2f dB -> denoised %.2f dB  (pose 0: %.2f -> %.2f)"
              % (tag, np.mean(raw), np.mean(den), raw[0], den[0]), flush=True)
        return np.mean(den)

    evaluate("before training")
    t0 = time.time()
    for it in range(args.iters):
        pick = rs.choice(train_idx, args.batch, replace=False)
        rz = rs.randint(0, 2, args.batch)
        aux = torch.stack([data[i][0][r] for i, r in zip(pick, rz)])
        noisy = torch.stack([data[i][1][r] for i, r in zip(pick, rz)])
        tgt = torch.stack([data[i][2] for i in pick])
        opt.zero_grad(set_to_none=True)
        out = denoiser.filtering(model, aux, noisy, requires_grad=True)
        loss = smape(out[..., :3], tgt[..., :3]) if args.loss == "smape" else ((out[..., :3] - tgt[..., :3]) ** 2).mean()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        sched.step()
        if it % 100 == 0 or it == args.iters - 1:
            print("it %4d loss %.5f  (%.1f it/s)" % (it, float(loss.detach()), (it + 1) / (time.time() - t0)), flush=True)
        if it % 500 == 499:
            evaluate("it %d" % (it + 1))
    evaluate("after training")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, args.out)
    print("saved", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
