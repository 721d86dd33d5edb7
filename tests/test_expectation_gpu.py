"""GPU twin of tests/test_expectation.py: the mean of 4096 spp rendered by the HIP kernels (batched
persistent kernel and the single-frame kernel, independently seeded launches) against the independent float64
rendering-equation model of tests/expected_render.py -- evidence that does not pass through oracle/."""
import numpy as np
import pytest

import expected_render as E
import rt_octree_amd as R
from rt_octree_amd import synth
from test_expectation import _check_against_model, _thin

pytestmark = pytest.mark.gpu


def test_hip_mean_matches_rendering_equation():
    t = _thin(synth.make_tree(depth_limit=6, basis_dim=9, seed=3, shell=1.5), 0.08)
    dt = R.N3Tree.from_arrays(t.child, t.data, t.scale, t.offset, t.data_format)
    W = H = 64
    fx = 0.9 * synth.blender_focal(W)
    pose = synth.orbit_poses(7)[3]
    cam = R.Camera(W, H, fx, fx)
    cam.set_c2w(pose)
    rot = [0.1, 0.3, -0.2]
    opt = R.RenderOptions(spp=32, denoise=False, background_brightness=0.5, rot_dirs=rot)
    # independently seeded launches (see _mean_of_frames: the reference's 2^32-strided frame streams are
    # correlated in the tails); two frames per launch of the batched kernel
    ctx = R.RenderContext(W, H, frames=2)
    acc = np.zeros((4, H, W))
    n_frames = 0
    for launch in range(64):
        ctx.rng_seed(977 + 7919 * launch)
        R.launch_renderer_batch(dt, [cam] * 2, opt, ctx, rng_jumps=[0, 1 + launch])
        for k in range(2):
            ctx.select_frame(k)
            acc += ctx.download_aux()[:4]
            n_frames += 1
    scene = E.Scene(t.child, t.data, t.scale, t.offset, t.data_format)
    mean, var = E.expected_frame(scene, pose, W, H, fx, fx, bg=0.5, rot_dirs=rot)
    _check_against_model(acc / n_frames, mean, var, n_frames * 32, "hip batched")
    # the single-frame kernel on other RNG streams
    one = R.RenderContext(W, H)
    acc[:] = 0
    for k in range(64):
        one.rng_seed(31337 + 104729 * k)
        R.launch_renderer(dt, cam, opt, one)
        acc += one.download_aux()[:4]
    _check_against_model(acc / 64, mean, var, 64 * 32, "hip single-frame")
