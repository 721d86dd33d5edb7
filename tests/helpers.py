"""Shared helpers for the parity tests: build the same scene for the oracle and for librto."""
import numpy as np

import orc
import rt_octree_amd as R
from rt_octree_amd import synth


def rgba_tree(tree_sh, seed=3):
    """An RGBA-format (data_dim 4, basis_dim -1) tree on the topology of `tree_sh`
    (rt_core.cuh:318-322 branch)."""
    rng = np.random.default_rng(seed)
    sig = tree_sh.data[..., -1:].astype(np.float32)
    rgb = rng.uniform(0, 1, tree_sh.data.shape[:-1] + (3,)).astype(np.float32) * (sig > 0)
    data = np.concatenate([rgb, sig], -1).astype(np.float16)
    return synth.SynthTree(tree_sh.child, data, tree_sh.scale, tree_sh.offset, "RGBA", tree_sh.depth_limit, {})


def make_pair(tree, device=0):
    """-> (orc.HostTree, R.N3Tree) over the same arrays"""
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format, device=device)
    return ht, dt


def cameras(W, H, pose, fx=None):
    fx = synth.blender_focal(W) if fx is None else fx
    cam = R.Camera(W, H, fx, fx)
    cam.set_c2w(pose)
    ocam = orc.camera(W, H, fx, fx, cam.transform.reshape(-1))
    return ocam, cam


def oracle_frame(ht, ocam, spp, frame=0, **optkw):
    opt = orc.default_options(spp=spp, **optkw)
    return orc.render_frame(ht, ocam, opt, orc.rng(frame=frame))


def hip_frame(dt, cam, spp, frame=0, kernel=R.KERNEL_AUTO, ctx=None, denoise=False, **optkw):
    ctx = ctx or R.RenderContext(cam.width, cam.height)
    ctx.rng_seed()
    for _ in range(frame):
        ctx.rng_advance()
    ctx.set_kernel(kernel)
    opt = R.RenderOptions(spp=spp, denoise=denoise, **optkw)
    R.launch_renderer(dt, cam, opt, ctx)
    return ctx.download_aux(), ctx.download_image(noisy=denoise), ctx


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    av, bv = a.view(np.uint32), b.view(np.uint32)
    bad = np.flatnonzero(av.reshape(-1) != bv.reshape(-1))
    assert bad.size == 0, "%s: %d of %d words differ; first at %d: %r vs %r" % (
        what, bad.size, av.size, bad[0], a.reshape(-1)[bad[0]], b.reshape(-1)[bad[0]])


def oracle_threads():
    """threads for whole-frame oracle renders: the cores this process may really use (affinity mask capped by the cgroup's CPU
    quota -- the GPU boxes show 256 CPUs behind a quota of 16, and 256 throttled OpenMP threads crawl)"""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def oracle_whole_frame(ht, cam, fx, spp, jump, fy=None):
    """orc.render_frame for one librto camera at RNG jump `jump` -> (aux [8,H,W], rgba [H,W,4])"""
    ocam = orc.camera(cam.width, cam.height, fx, fx if fy is None else fy, cam.transform.reshape(-1))
    aux, rgba, _ = orc.render_frame(ht, ocam, orc.default_options(spp=spp), orc.rng(frame=jump), threads=oracle_threads(), want_stats=False)
    return aux, rgba
