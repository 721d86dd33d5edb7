"""HIP render path (through the C ABI) vs the CPU oracle: bit-exact fp32 aux planes, RGBA32F image
and RGBA8 bytes on the same seeded inputs.  Tolerance: none (0 ulp) -- both sides evaluate the same
IEEE expression sequence with the shared deterministic logf/expf definition (DESIGN.md "Math")."""
import os

import numpy as np
import pytest

import orc
import rt_octree_amd as R
from helpers import assert_bits_equal, cameras, hip_frame, make_pair, oracle_frame, rgba_tree
from rt_octree_amd import synth

pytestmark = pytest.mark.gpu

POSES = synth.orbit_poses(5)


@pytest.mark.parametrize("spp", R.SUPPORTED_SPP)
@pytest.mark.parametrize("kernel", [R.KERNEL_GENERIC, R.KERNEL_FAST])
def test_sh9_all_spp_bit_exact(small_tree_sh9, spp, kernel):
    ht, dt = make_pair(small_tree_sh9)
    W, H = 96, 80  # ragged vs the 32x8 tiles
    ocam, cam = cameras(W, H, POSES[1])
    aux_o, rgba_o, st = oracle_frame(ht, ocam, spp, frame=3)
    aux_h, rgba_h, ctx = hip_frame(dt, cam, spp, frame=3, kernel=kernel)
    assert st["hit_rays"] > 100
    assert_bits_equal(aux_h, aux_o, "aux spp=%d" % spp)
    assert_bits_equal(rgba_h, rgba_o, "rgba spp=%d" % spp)
    assert np.array_equal(ctx.download_rgba8(), orc.rgba8(rgba_o))


@pytest.mark.parametrize("kernel", [R.KERNEL_GENERIC, R.KERNEL_FAST])
@pytest.mark.parametrize("pose", range(len(POSES)))
def test_sh16_poses_bit_exact(small_tree_sh16, pose, kernel):
    ht, dt = make_pair(small_tree_sh16)
    W, H = 100, 60
    ocam, cam = cameras(W, H, POSES[pose])
    aux_o, rgba_o, _ = oracle_frame(ht, ocam, 6, frame=pose)
    aux_h, rgba_h, _ = hip_frame(dt, cam, 6, frame=pose, kernel=kernel)
    assert_bits_equal(aux_h, aux_o, "aux")
    assert_bits_equal(rgba_h, rgba_o, "rgba")


@pytest.mark.parametrize("kernel", [R.KERNEL_GENERIC, R.KERNEL_FAST])
def test_rgba_format_bit_exact(small_tree_sh9, kernel):
    tree = rgba_tree(small_tree_sh9)
    ht, dt = make_pair(tree)
    ocam, cam = cameras(64, 64, POSES[2])
    aux_o, rgba_o, _ = oracle_frame(ht, ocam, 4)
    aux_h, rgba_h, _ = hip_frame(dt, cam, 4, kernel=kernel)
    assert_bits_equal(aux_h, aux_o, "aux")
    assert_bits_equal(rgba_h, rgba_o, "rgba")


def test_options_paths_bit_exact(small_tree_sh9):
    """render_bbox crop, basis_minmax mask, background, sigma_thresh, step_size."""
    ht, dt = make_pair(small_tree_sh9)
    ocam, cam = cameras(72, 72, POSES[0])
    kw = dict(render_bbox=[0.1, 0.2, 0.0, 0.9, 0.8, 0.7], basis_minmax=[1, 5], background_brightness=0.25,
              sigma_thresh=8.0, step_size=3e-4)
    aux_o, rgba_o, _ = oracle_frame(ht, ocam, 3, **kw)
    for kernel in (R.KERNEL_GENERIC, R.KERNEL_FAST):
        aux_h, rgba_h, _ = hip_frame(dt, cam, 3, kernel=kernel, **kw)
        assert_bits_equal(aux_h, aux_o, "aux")
        assert_bits_equal(rgba_h, rgba_o, "rgba")


@pytest.mark.parametrize("rot", [(0.3, -0.2, 0.5), (0.0, 0.0, 0.7853982), (1e-7, 0.0, 0.0)])
def test_rot_dirs_rodrigues_bit_exact(small_tree_sh16, rot):
    """opt.rot_dirs turns the VIEW direction of the SH lookup (rodrigues, volrend.cu:58-73,155); the
    marching direction is untouched.  Generic, fast and batched kernels vs the oracle; the last case
    lies below the 1e-6 cut-off and must equal the unrotated frame."""
    ht, dt = make_pair(small_tree_sh16)
    ocam, cam = cameras(64, 56, POSES[2])
    aux_o, rgba_o, _ = oracle_frame(ht, ocam, 4, rot_dirs=list(rot))
    aux_0, _, _ = oracle_frame(ht, ocam, 4)
    if rot[0] == 1e-7:
        assert_bits_equal(aux_o, aux_0, "below the cut-off")
    else:
        assert not np.array_equal(aux_o[:3], aux_0[:3])  # colours move ...
        assert_bits_equal(aux_o[3], aux_0[3], "alpha")     # ... the traversal does not
    for kernel in (R.KERNEL_GENERIC, R.KERNEL_FAST):
        aux_h, rgba_h, _ = hip_frame(dt, cam, 4, kernel=kernel, rot_dirs=list(rot))
        assert_bits_equal(aux_h, aux_o, "aux")
        assert_bits_equal(rgba_h, rgba_o, "rgba")
    ctx = R.RenderContext(64, 56, frames=2)
    R.launch_renderer_batch(dt, [cam, cam], R.RenderOptions(spp=4, denoise=False, rot_dirs=list(rot)), ctx, rng_jumps=[7, 0])
    ctx.select_frame(1)
    assert_bits_equal(ctx.download_aux(), aux_o, "batched aux")


def test_node_order_never_changes_pixels(small_tree_sh9):
    """The same octree with its nodes stored in a random order (synth.shuffle_nodes) renders the same
    image through every kernel: node order is a file-layout accident, not scene content."""
    shuffled = synth.shuffle_nodes(small_tree_sh9, seed=5)
    ht, dt = make_pair(small_tree_sh9)
    _, dts = make_pair(shuffled)
    ocam, cam = cameras(72, 64, POSES[1])
    aux_o, rgba_o, _ = oracle_frame(ht, ocam, 6)
    for kernel in (R.KERNEL_GENERIC, R.KERNEL_FAST):
        aux_h, rgba_h, _ = hip_frame(dts, cam, 6, kernel=kernel)
        assert_bits_equal(aux_h, aux_o, "aux")
        assert_bits_equal(rgba_h, rgba_o, "rgba")
    ctx = R.RenderContext(72, 64, frames=1)
    R.launch_renderer_batch(dts, [cam], R.RenderOptions(spp=6, denoise=False), ctx)
    assert_bits_equal(ctx.download_aux(), aux_o, "batched aux")


def test_batched_launch_on_a_tree_without_traversal_image():
    """N = 3 (any tree the N == 2 traversal image cannot encode): rto_launch_renderer_batch renders the
    same frames through per-frame launches of the generic kernel instead of refusing."""
    rs = np.random.RandomState(9)
    child = np.zeros((3, 3, 3, 3), np.int32)
    child[0, 1, 1, 1] = 1  # node 1 = centre cell of the root
    child[0, 2, 0, 1] = 2
    data = np.zeros((3, 3, 3, 3, 4), np.float16)
    data[..., :3] = rs.uniform(0, 1, (3, 3, 3, 3, 3))
    data[..., 3] = rs.uniform(0, 12, (3, 3, 3, 3)) * (rs.uniform(size=(3, 3, 3, 3)) < 0.6)
    t = synth.SynthTree(child, data, np.full(3, 1 / 3.0, np.float32), np.full(3, 0.5, np.float32), "RGBA", 2, {})
    ht, dt = make_pair(t)
    assert dt.N == 3
    cams, ocams = [], []
    for i in range(3):
        ocam, cam = cameras(40, 32, POSES[i])
        cams.append(cam)
        ocams.append(ocam)
    ctx = R.RenderContext(40, 32, frames=3)
    R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=4, denoise=False), ctx, rng_jumps=[2, 0, 5])
    for i, j in enumerate([2, 0, 5]):
        aux_o, rgba_o, _ = oracle_frame(ht, ocams[i], 4, frame=j)
        ctx.select_frame(i)
        assert_bits_equal(ctx.download_aux(), aux_o, "aux %d" % i)
        assert_bits_equal(ctx.download_image(), rgba_o, "rgba %d" % i)
    assert aux_o[3].max() > 0


@pytest.mark.parametrize("hook", [1, 2])
def test_batched_fallback_on_a_tree_that_released_its_reference_arrays(hook):
    """ADVICE r3 (high + medium): the per-frame generic fallback inside rto_launch_renderer_batch -- taken for a tree with
    more leaf slots than a hit entry can name at the SPP (hook 1) or when the device refuses the traversal kernel's LDS
    (hook 2) -- on a dense SH tree whose upload RELEASED child[] / data[] (the default): the arrays are rebuilt first
    (round 3 launched the generic kernel on null pointers), the frames equal the oracle's, and the tile marks of an
    EARLIER batched launch on the context are gone (the denoise stage would fill "culled" tiles of the new frames)."""
    tree = synth.make_tree(depth_limit=6, basis_dim=9, seed=77)
    ht, dt = make_pair(tree)
    before = dt.device_bytes
    cams, ocams = [], []
    for i in range(3):
        ocam, cam = cameras(64, 48, POSES[i])
        cams.append(cam)
        ocams.append(ocam)
    ctx = R.RenderContext(64, 48, frames=3)
    opt = R.RenderOptions(spp=4, denoise=False)
    R.launch_renderer_batch(dt, cams, opt, ctx, rng_jumps=[2, 0, 5])  # the batched kernels: marks exist
    assert ctx.tile_marks() is not None and ctx.tile_marks()[3] == 3
    dt._refresh()
    assert dt.device_bytes == before
    fast = []
    for i in range(3):
        ctx.select_frame(i)
        fast.append(ctx.download_aux())
    ctx.set_tuning("batch_fallback", hook)
    R.launch_renderer_batch(dt, cams, opt, ctx, rng_jumps=[2, 0, 5])
    assert ctx.tile_marks() is None  # stale marks must not survive
    dt._refresh()
    assert dt.device_bytes >= before + tree.data.nbytes + tree.child.nbytes  # the generic kernel's arrays came back
    for i, j in enumerate([2, 0, 5]):
        aux_o, rgba_o, _ = oracle_frame(ht, ocams[i], 4, frame=j)
        ctx.select_frame(i)
        assert_bits_equal(ctx.download_aux(), aux_o, "fallback aux %d" % i)
        assert_bits_equal(ctx.download_image(), rgba_o, "fallback rgba %d" % i)
        assert_bits_equal(fast[i], aux_o, "batched aux %d" % i)
    ctx.set_tuning("batch_fallback", 0)
    R.launch_renderer_batch(dt, cams, opt, ctx, rng_jumps=[2, 0, 5])
    assert ctx.tile_marks() is not None


def _chain_tree(depth, seed, basis=4):
    """a hand-built deep, narrow octree: every node refines ONE child (towards a corner that moves with the level), the other
    seven are leaves of random density -- `depth` node levels with 8 * depth slots: the two-level image then has pairs well
    beyond the two of the bench tree, at both parities of (depth - grid levels)"""
    rs = np.random.RandomState(seed)
    child = np.zeros((depth, 2, 2, 2), np.int32)
    dd = 3 * basis + 1
    data = np.zeros((depth, 2, 2, 2, dd), np.float16)
    data[..., :dd - 1] = rs.normal(0, 0.4, (depth, 2, 2, 2, dd - 1))
    data[..., dd - 1] = rs.uniform(0, 40, (depth, 2, 2, 2)) * (rs.uniform(size=(depth, 2, 2, 2)) < 0.7)
    for lvl in range(depth - 1):
        a, b, c = (lvl * 5 + seed) & 1, (lvl * 3 + (seed >> 1)) & 1, (lvl + (seed >> 2)) & 1
        child[lvl, a, b, c] = 1  # the next node in breadth-first order
    return synth.SynthTree(child, data, np.full(3, 1 / 3.0, np.float32), np.full(3, 0.5, np.float32), "SH%d" % basis, depth, {})


@pytest.mark.parametrize("depth", [9, 12, 13, 17, 24])
def test_deep_narrow_trees_walk_the_two_level_image_bit_exact(depth):
    """depths up to the 24 levels the fixed-point coordinates allow: up to 9 pairs of levels below the grid, odd and even;
    batched and single-frame kernels against the oracle, and against the same tree without the two-level image"""
    tree = _chain_tree(depth, seed=depth)
    ht, dt = make_pair(tree)
    assert dt.max_depth == depth and dt.wide_nodes == len(range(6, depth, 2))
    cams, ocams = [], []
    for i, pos in enumerate([(2.2, 1.7, 1.9), (-1.9, 2.4, 0.8), (0.4, 0.3, 2.9)]):
        ocam, cam = cameras(56, 40, synth.look_at_c2w(pos, target=(0.1 * i, -0.1, 0.05)))
        cams.append(cam)
        ocams.append(ocam)
    ctx = R.RenderContext(56, 40, frames=3)
    opt = R.RenderOptions(spp=8, denoise=False)
    R.launch_renderer_batch(dt, cams, opt, ctx, rng_jumps=[3, 4, 5])
    for i in range(3):
        aux_o, rgba_o, _ = oracle_frame(ht, ocams[i], 8, frame=3 + i)
        ctx.select_frame(i)
        assert_bits_equal(ctx.download_aux(), aux_o, "batched aux %d" % i)
        assert_bits_equal(ctx.download_image(), rgba_o, "batched image %d" % i)
        assert aux_o[3].max() > 0
    aux_f, _, _ = hip_frame(dt, cams[0], 8, frame=3, kernel=R.KERNEL_FAST)
    assert_bits_equal(aux_f, oracle_frame(ht, ocams[0], 8, frame=3)[0], "single-frame kernel")


@pytest.mark.parametrize("depth,basis", [(1, 4), (2, 4), (3, 9), (8, 9)])
def test_batched_path_on_shallow_and_deep_trees(depth, basis):
    """no top grid (depth < 3), a shallow grid, a deeper tree; SPP 1 and 32; a negative sigma_thresh makes even
    zero-density leaves candidates for a hit"""
    tree = synth.make_tree(depth_limit=depth, basis_dim=basis, seed=depth)
    ht, dt = make_pair(tree)
    ocam, cam = cameras(56, 40, POSES[3])
    ctx = R.RenderContext(56, 40, frames=2)
    for spp, kw in ((1, {}), (32, {}), (4, {"sigma_thresh": -1.0})):
        want = oracle_frame(ht, ocam, spp, frame=1, **kw)[0]
        ctx.rng_seed()
        R.launch_renderer_batch(dt, [cam, cam], R.RenderOptions(spp=spp, denoise=False, **kw), ctx, rng_jumps=[4, 1])
        ctx.select_frame(1)
        assert_bits_equal(ctx.download_aux(), want, "depth %d spp %d" % (depth, spp))


@pytest.mark.parametrize("basis", [9, 16])
def test_shading_from_the_aligned_copy_and_from_data_agree(basis):
    """dense SH9 / SH16 trees are shaded from an aligned copy of their coefficients (TreeDev::shrec); with the copy
    switched off (RTO_TREE_COMPACT) the kernels read the 2-byte-aligned records of data[] -- same fp16 values, same
    pixels.  With the copy the reference-layout arrays are released after the upload (RTO_TREE_KEEP_REFERENCE keeps
    them): the footprint is the padded copy + the traversal image, not data + copy."""
    tree = synth.make_tree(depth_limit=5, basis_dim=basis, seed=basis)
    ht, dt = make_pair(tree)
    dt_keep = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format, keep_reference=True)
    dt_plain = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format, compact=True)
    assert dt_keep.device_bytes > tree.data.nbytes * 1.9  # data + the padded copy
    assert dt.device_bytes < dt_keep.device_bytes - tree.data.nbytes * 0.99  # data[] and child[] released
    assert dt_plain.device_bytes < dt_keep.device_bytes
    ocam, cam = cameras(64, 48, POSES[1])
    want = oracle_frame(ht, ocam, 6, frame=2)[0]
    for t in (dt, dt_keep, dt_plain):
        assert_bits_equal(hip_frame(t, cam, 6, frame=2, kernel=R.KERNEL_FAST)[0], want, "single frame")
        ctx = R.RenderContext(64, 48, frames=1)
        R.launch_renderer_batch(t, [cam], R.RenderOptions(spp=6, denoise=False), ctx, rng_jumps=[2])
        assert_bits_equal(ctx.download_aux(), want, "batched")


@pytest.mark.parametrize("basis", [9, 16])
def test_compact_coefficient_records_render_the_same_pixels(basis, tmp_path):
    """RTO_TREE_COMPACT_RECORDS: records only for the leaf slots of positive density, found through a per-slot index --
    every kernel (fast, batched, generic after the rebuild) renders the oracle's pixels from half the resident bytes;
    a negative sigma_thresh, under which a leaf without a record could be hit, is refused"""
    tree = synth.make_tree(depth_limit=6, basis_dim=basis, seed=60 + basis)
    ht, dt = make_pair(tree)
    dc = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format, compact_records=True)
    occupied = int(((tree.child.reshape(-1) == 0) & (tree.data[..., -1].reshape(-1).astype(np.float32) > 0)).sum())
    rec_bytes = 64 if basis == 9 else 128
    # round 5: the default layout keeps one record per ENTRY of the two-level image (a hit names its record directly): the wide
    # nodes' 64 entries each + the top-grid cells padded to whole nodes; the compact layout one per occupied slot + the index
    G = min(dt.max_depth - 1, 6)
    n_records = (dt.wide_nodes + ((8 ** G + 63) // 64)) * 64 if dt.wide_nodes else tree.child.size
    assert dt.device_bytes - dc.device_bytes == (n_records - max(occupied, 1)) * rec_bytes - tree.child.size * 4
    assert occupied < 0.7 * tree.child.size  # (the scene is mostly empty space: that is the point)
    ocam, cam = cameras(80, 56, POSES[3])
    want = oracle_frame(ht, ocam, 6, frame=4)
    for kernel in (R.KERNEL_FAST, R.KERNEL_GENERIC):
        aux, rgba, _ = hip_frame(dc, cam, 6, frame=4, kernel=kernel)
        assert_bits_equal(aux, want[0], "aux kernel %d" % kernel)
        assert_bits_equal(rgba, want[1], "rgba kernel %d" % kernel)
    ctx = R.RenderContext(80, 56, frames=2)
    R.launch_renderer_batch(dc, [cam, cam], R.RenderOptions(spp=6, denoise=False), ctx, rng_jumps=[4, 4])
    for f in range(2):
        ctx.select_frame(f)
        assert_bits_equal(ctx.download_aux(), want[0], "batched slot %d" % f)
    for launch in (lambda o: R.launch_renderer(dc, cam, o, ctx), lambda o: R.launch_renderer_batch(dc, [cam], o, ctx)):
        with pytest.raises(R.RtoError) as e:
            launch(R.RenderOptions(spp=6, denoise=False, sigma_thresh=-1.0))
        assert "sigma_thresh must be >= 0" in str(e.value)
    # from a file, with the flag of rto_tree_load_npz_ex
    p = str(tmp_path / "t.npz")
    tree.save_npz(p)
    df = R.N3Tree(p, compact_records=True)
    assert df.device_bytes == dc.device_bytes
    assert_bits_equal(hip_frame(df, cam, 6, frame=4)[0], want[0], "from the file")


@pytest.mark.parametrize("basis", [9, 16])
def test_generic_kernel_on_a_tree_without_reference_arrays(basis):
    """the generic kernel (root-restart float descent over child[] / data[], the plain statement of the reference) on a
    tree whose upload released those arrays: they are rebuilt from the traversal image + the aligned copy on first
    use -- same leaf values, so the oracle's pixels -- and the footprint grows by exactly what was released"""
    tree = synth.make_tree(depth_limit=6, basis_dim=basis, seed=40 + basis)
    ht, dt = make_pair(tree)
    before = dt.device_bytes
    ocam, cam = cameras(72, 56, POSES[2])
    want = oracle_frame(ht, ocam, 4, frame=3)
    aux_f, rgba_f, ctx = hip_frame(dt, cam, 4, frame=3, kernel=R.KERNEL_FAST)
    dt._refresh()
    assert dt.device_bytes == before  # the fast kernel needed nothing back
    aux_g, rgba_g, _ = hip_frame(dt, cam, 4, frame=3, kernel=R.KERNEL_GENERIC, ctx=ctx)
    dt._refresh()
    assert dt.device_bytes >= before + tree.data.nbytes + tree.child.nbytes
    for a in (aux_f, aux_g):
        assert_bits_equal(a, want[0], "aux")
    assert_bits_equal(rgba_g, want[1], "rgba")
    aux_f2, _, _ = hip_frame(dt, cam, 4, frame=3, kernel=R.KERNEL_FAST, ctx=ctx)  # and the fast kernel still renders
    assert_bits_equal(aux_f2, want[0], "fast after the rebuild")


def test_camera_inside_box_and_miss(small_tree_sh9):
    """tmin clamps at 0 for a camera inside the volume; rays that miss return background
    (rt_core.cuh:219-222; SURVEY appendix B 3,4)."""
    ht, dt = make_pair(small_tree_sh9)
    inside = synth.look_at_c2w((0.3, 0.2, 0.4), target=(0, 0, -0.2))
    away = synth.look_at_c2w((4, 0, 0), target=(8, 0, 0))
    for pose in (inside, away):
        ocam, cam = cameras(48, 40, pose)
        aux_o, rgba_o, _ = oracle_frame(ht, ocam, 2)
        for kernel in (R.KERNEL_GENERIC, R.KERNEL_FAST):
            aux_h, rgba_h, _ = hip_frame(dt, cam, 2, kernel=kernel)
            assert_bits_equal(aux_h, aux_o, "aux")
            assert_bits_equal(rgba_h, rgba_o, "rgba")
    assert np.all(aux_h[3] == 0) and np.all(rgba_h[..., :3] == 1.0)


def test_denoise_flag_selects_noisy_target(small_tree_sh9):
    """opt.denoise routes the image write to the noisy buffer (volrend.cu:206)."""
    _, dt = make_pair(small_tree_sh9)
    _, cam = cameras(32, 32, POSES[0])
    ctx = R.RenderContext(32, 32)
    aux1, img_final, _ = hip_frame(dt, cam, 1, ctx=ctx, denoise=False)
    aux2, img_noisy, _ = hip_frame(dt, cam, 1, ctx=ctx, denoise=True)
    assert_bits_equal(aux1, aux2)
    assert_bits_equal(img_final, img_noisy)


def test_unsupported_spp_raises(small_tree_sh9):
    _, dt = make_pair(small_tree_sh9)
    _, cam = cameras(16, 16, POSES[0])
    ctx = R.RenderContext(16, 16)
    with pytest.raises(R.RtoError) as e:
        R.launch_renderer(dt, cam, R.RenderOptions(spp=5), ctx)
    assert "spp == 5 not supported" in str(e.value)


def test_full_size_properties(small_tree_sh9):
    """800x800 SPP 6 (BASELINE config size): size-independent properties + generic == fast."""
    tree = synth.make_tree(depth_limit=8, basis_dim=9, seed=5)
    _, dt = make_pair(tree)
    _, cam = cameras(800, 800, POSES[3])
    aux_f, rgba_f, ctx = hip_frame(dt, cam, 6, frame=100, kernel=R.KERNEL_FAST)
    aux_g, rgba_g, _ = hip_frame(dt, cam, 6, frame=100, kernel=R.KERNEL_GENERIC)
    assert_bits_equal(aux_f, aux_g, "fast vs generic aux")
    assert_bits_equal(rgba_f, rgba_g, "fast vs generic rgba")
    alpha = aux_f[3]
    assert np.all(np.isin(np.round(alpha * 6).astype(int), range(7))) and np.all(np.abs(alpha * 6 - np.round(alpha * 6)) < 1e-5)
    assert np.all(aux_f[4:] == aux_f[:4] ** 2)                     # squares planes
    assert np.all(rgba_f[..., 3] == 1.0)                           # alpha forced to 1
    assert np.all((rgba_f[..., :3] >= 0) & (rgba_f[..., :3] <= 1.0 + 1e-6))
    assert np.all(rgba_f[..., :3][alpha == 0] == 1.0)              # misses = background
    aux_r, _, _ = hip_frame(dt, cam, 6, frame=100, kernel=R.KERNEL_FAST, ctx=ctx)
    assert_bits_equal(aux_r, aux_f, "re-render determinism")
    aux_n, _, _ = hip_frame(dt, cam, 6, frame=101, kernel=R.KERNEL_FAST, ctx=ctx)
    assert not np.array_equal(aux_n, aux_f)                        # per-frame RNG jump changes the noise
    # spot-check 64 pixels of the full-size frame against the oracle
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    ocam, _ = cameras(800, 800, POSES[3])
    import ctypes as C
    opt = orc.default_options(spp=6)
    base = orc.rng(frame=100)
    rs = np.random.RandomState(0)
    for idx in rs.randint(0, 640000, 64):
        a8 = (C.c_float * 8)()
        px = (C.c_float * 4)()
        rc = orc.lib().orc_render_pixel(C.byref(ht.c), C.byref(ocam), C.byref(opt), C.byref(base), int(idx), a8, px, None)
        assert rc == 0
        got = aux_f[:, idx // 800, idx % 800]
        assert np.array_equal(np.array(a8[:], np.float32).view(np.uint32), got.view(np.uint32)), idx


def test_hip_matches_committed_golden_frames():
    """HIP path vs tests/golden/frames_golden.npz (aux fp32 bits + RGBA8 bytes) -- frames produced by this repository's
    oracle (a self-regression pin, not reference output)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames_golden.npz"))
    W, H, fx = g["size_fx"]
    W, H = int(W), int(H)
    for name, fmt in (("sh9", "SH9"), ("sh16", "SH16")):
        dt = R.N3Tree.from_arrays(g[name + ".child"], g[name + ".data"], g[name + ".scale"], g[name + ".offset"], fmt)
        ctx = R.RenderContext(W, H)
        for spp in (1, 6):
            for pi in range(3):
                _, cam = cameras(W, H, g["poses"][pi], fx=float(fx))
                for kernel in (R.KERNEL_GENERIC, R.KERNEL_FAST):
                    aux, _, _ = hip_frame(dt, cam, spp, frame=100 + pi, kernel=kernel, ctx=ctx)
                    key = "%s.spp%d.pose%d" % (name, spp, pi)
                    assert_bits_equal(aux, g[key + ".aux"], key)
                    assert np.array_equal(ctx.download_rgba8(), g[key + ".rgba8"]), key


def test_work_counters_match_oracle(small_tree_sh16):
    """the units behind bench.py's algorithmic-byte count are the oracle's own counters"""
    ht, dt = make_pair(small_tree_sh16)
    ocam, cam = cameras(96, 64, POSES[4])
    _, _, st = oracle_frame(ht, ocam, 6, frame=1)
    ctx = R.RenderContext(96, 64)
    ctx.enable_stats(True)
    aux_s, _, _ = hip_frame(dt, cam, 6, frame=1, kernel=R.KERNEL_FAST, ctx=ctx)
    assert ctx.get_stats() == st
    ctx.enable_stats(False)
    aux_t, _, _ = hip_frame(dt, cam, 6, frame=1, kernel=R.KERNEL_FAST, ctx=ctx)
    assert_bits_equal(aux_s, aux_t, "counting instantiation renders the same frame")


def test_compact_flag_skips_the_shading_copy(tmp_path, small_tree_sh16):
    p = str(tmp_path / "t.npz")
    small_tree_sh16.save_npz(p)
    full, compact = R.N3Tree(p), R.N3Tree(p, compact=True)
    assert compact.device_bytes < full.device_bytes
    _, cam = cameras(48, 40, POSES[0])
    a = hip_frame(full, cam, 6)[0]
    b = hip_frame(compact, cam, 6)[0]
    assert_bits_equal(a, b, "compact vs default")


def test_tree_npz_roundtrip_on_device(tmp_path, small_tree_sh9):
    """tree.npz (svox schema) -> rto_tree_load_npz renders the same bits as the in-memory upload"""
    p = str(tmp_path / "tree.npz")
    small_tree_sh9.save_npz(p, compressed=True)
    dt_file = R.N3Tree(p)
    assert (dt_file.capacity, dt_file.N, dt_file.data_dim, dt_file.data_format) == (small_tree_sh9.capacity, 2, 28, "SH9")
    _, dt_mem = make_pair(small_tree_sh9)
    _, cam = cameras(64, 40, POSES[1])
    a1, _, _ = hip_frame(dt_file, cam, 4)
    a2, _, _ = hip_frame(dt_mem, cam, 4)
    assert_bits_equal(a1, a2)
    with pytest.raises(R.RtoError):
        R.N3Tree(str(tmp_path / "missing.npz"))


def test_ndc_path_bit_exact(small_tree_sh9):
    """LLFF NDC warp (maybe_world2ndc volrend.cu:35-56)."""
    t = small_tree_sh9
    ht = orc.HostTree(t.child, t.data, t.scale, t.offset, t.data_format, ndc=(64.0, 48.0, 50.0))
    dt = R.N3Tree.from_arrays(t.child, t.data, t.scale, t.offset, t.data_format)
    dt.set_ndc(64.0, 48.0, 50.0)
    pose = synth.look_at_c2w((0.1, 0.05, 0.3), target=(0, 0, -1), up=(0, 1, 0))
    ocam, cam = cameras(64, 48, pose, fx=50.0)
    aux_o, rgba_o, st = oracle_frame(ht, ocam, 2)
    for kernel in (R.KERNEL_GENERIC, R.KERNEL_FAST):
        aux_h, rgba_h, _ = hip_frame(dt, cam, 2, kernel=kernel)
        assert_bits_equal(aux_h, aux_o, "ndc aux")


@pytest.mark.parametrize("spp", [1, 6, 32])
def test_batched_persistent_kernel_bit_exact(small_tree_sh16, spp):
    """rto_launch_renderer_batch (persistent ray-queue kernel, several frames per launch) ==
    the reference's frame loop: launch_renderer; ctx.rng.advance()."""
    ht, dt = make_pair(small_tree_sh16)
    W, H = 100, 52  # ragged vs the 8x8 ray tiles
    cams, want = [], []
    for f in range(5):
        ocam, cam = cameras(W, H, POSES[f])
        cams.append(cam)
        aux_o, rgba_o, _ = oracle_frame(ht, ocam, spp, frame=7 + f)
        want.append((aux_o, rgba_o))
    ctx = R.RenderContext(W, H, frames=5)
    ctx.rng_seed()
    for _ in range(7):
        ctx.rng_advance()
    for rep in range(2):  # second launch exercises the queue re-arm
        R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=spp, denoise=False), ctx)
        for f in range(5):
            ctx.select_frame(f)
            assert_bits_equal(ctx.download_aux(), want[f][0], "aux frame %d" % f)
            assert_bits_equal(ctx.download_image(), want[f][1], "rgba frame %d" % f)
    # explicit jump counts: frame slots in any order
    R.launch_renderer_batch(dt, [cams[3], cams[1]], R.RenderOptions(spp=spp, denoise=True), ctx, rng_jumps=[3, 1])
    for slot, f in ((0, 3), (1, 1)):
        ctx.select_frame(slot)
        assert_bits_equal(ctx.download_image(noisy=True), want[f][1], "noisy slot %d" % slot)


def test_batched_kernel_full_size_matches_fast(small_tree_sh9):
    tree = synth.make_tree(depth_limit=8, basis_dim=9, seed=5)
    _, dt = make_pair(tree)
    cams = []
    for f in range(3):
        _, cam = cameras(800, 800, POSES[f])
        cams.append(cam)
    ctx = R.RenderContext(800, 800, frames=3)
    R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=False), ctx)
    got = []
    for f in range(3):
        ctx.select_frame(f)
        got.append(ctx.download_aux())
    ctx.select_frame(0)
    for f in range(3):
        aux_f, _, _ = hip_frame(dt, cams[f], 6, frame=f, kernel=R.KERNEL_FAST)
        assert_bits_equal(got[f], aux_f, "frame %d" % f)


def test_queue_tuning_never_changes_results(small_tree_sh16):
    """rto_ctx_set_tuning: single queue / one queue per XCD, frame- / tile-major order, row-major /
    centre-out / wedge / band tile tables of any block size, refill thresholds -- which wave renders which ray
    when is free, the pixels are not."""
    ht, dt = make_pair(small_tree_sh16)
    W, H = 132, 76  # ragged vs the 8x8 ray tiles and vs the 4x4-tile blocks
    cams, want = [], []
    for f in range(3):
        ocam, cam = cameras(W, H, POSES[f])
        cams.append(cam)
        want.append(oracle_frame(ht, ocam, 6, frame=f)[0])
    ctx = R.RenderContext(W, H, frames=3)
    settings = [{"xcd_queues": 0, "tile_order": 0}, {"xcd_queues": 0, "tile_order": 1, "tile_major": 0},
                {"xcd_queues": 1, "tile_major": 1, "tile_block": 1}, {"tile_block": 3}, {"tile_block": 64},
                {"tile_block": 4, "refill": 808}, {"refill": 816}, {"refill": 432}, {"refill": 0},
                {"queue_bands": 0}, {"queue_bands": 1}, {"queue_bands": 3}, {"queue_bands": 64}]  # angular wedges / bands of tile rows per XCD queue
    for kv in settings:
        for k, v in kv.items():
            ctx.set_tuning(k, v)
        ctx.rng_seed()
        R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=False), ctx)
        for f in range(3):
            ctx.select_frame(f)
            assert_bits_equal(ctx.download_aux(), want[f], "%s frame %d" % (kv, f))
    with pytest.raises(R.RtoError):
        ctx.set_tuning("tile_block", 0)
    with pytest.raises(R.RtoError):
        ctx.set_tuning("queue_bands", 65)
    with pytest.raises(R.RtoError):
        ctx.set_tuning("no_such_knob", 1)


def test_batched_launches_repeat_bit_for_bit(small_tree_sh16):
    """work stealing and refill order differ from launch to launch; the frames must not: 8 launches of
    8 frames, every byte of every output hashed"""
    import hashlib
    _, dt = make_pair(small_tree_sh16)
    W, H = 320, 240
    cams = []
    for f in range(8):
        _, cam = cameras(W, H, POSES[f % len(POSES)])
        cams.append(cam)
    ctx = R.RenderContext(W, H, frames=8)
    seen = set()
    for rep in range(8):
        ctx.rng_seed()
        R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=6, denoise=False), ctx, rng_jumps=list(range(50, 58)))
        h = hashlib.sha1()
        for f in range(8):
            ctx.select_frame(f)
            h.update(ctx.download_aux().tobytes())
            h.update(ctx.download_image().tobytes())
        seen.add(h.hexdigest())
    assert len(seen) == 1


@pytest.mark.skipif(bool(os.environ.get("PYTEST_XDIST_WORKER")), reason="free-memory accounting needs the GPU to itself")
def test_handles_release_their_device_memory(small_tree_sh9):
    """create / use / free trees, contexts (all lazily grown buffers included) and fused networks in a
    loop: free device memory returns to where it started"""
    import torch
    from rt_octree_amd import denoiser
    tree = small_tree_sh9
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    _, cam = cameras(256, 192, POSES[0])

    def cycle():
        dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
        ctx = R.RenderContext(256, 192, frames=4)
        R.launch_renderer_batch(dt, [cam] * 4, R.RenderOptions(spp=8, denoise=True), ctx)
        R.launch_renderer(dt, cam, R.RenderOptions(spp=2, denoise=False), ctx)
        ctx.download_aux()
        ctx.freeResource()  # RenderContext::freeResource (render_context.hpp)
        dt.free()

    cycle()  # first use may grow allocator pools / load code objects
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), (free0, free1)


@pytest.mark.gpu
def test_every_threshold_draw_matches_the_oracle():
    """sample_dst's draws (rt_core.cuh:67-88: t = -logf(1 - rng.next_float())) can take 2^23 values, one per RNG float
    k / 2^23.  The device computes them with a shorter instruction sequence than the oracle's det_logf (reciprocal +
    Newton instead of the IEEE division, fused multiply-adds): every one of the 2^23 results must still be the
    oracle's float, bit for bit."""
    import ctypes as C

    from rt_octree_amd._lib import check, lib
    n = 1 << 23
    got = np.empty(n, np.float32)
    check(lib().rto_probe_thresholds(0, n, got.ctypes.data_as(C.c_void_p)))
    want = np.empty(n, np.float32)
    orc.lib().orc_thresholds(0, n, want.ctypes.data_as(C.c_void_p))
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, "first mismatching draws k = %s" % bad[:8]
    assert got[0] == 0.0 and np.all(got[1:] > 0) and np.all(np.isfinite(got))
    assert np.all(np.diff(got.astype(np.float64)) >= 0)  # -log(1 - u) is monotone in u


def test_short_sigmoid_equals_the_plain_statement_for_every_float_and_sample_count():
    """round 6: the shading kernels evaluate `cnt / (1.f + expf(-t))` (rt_core.cuh:314-318) through sigmoid_cnt3 -- det_expf
    for |t| <= 87 with fused multiply-adds, the division as v_rcp_f32 + one exact-residual correction.  Same float as the plain
    statement (whose det_expf the sweep below pins to the oracle) for ALL 2^32 floats t times every sample count 1 .. 32, and
    the short division alone for every float d in [1, 2^126) times every count: exhaustive, on the device (v_rcp_f32's bits
    are the hardware's)."""
    import ctypes as C

    from rt_octree_amd._lib import check, lib
    out = (C.c_uint64 * 3)()
    pairs = 0
    for chunk in range(8):  # mode 0: all floats, 2^29 per launch
        check(lib().rto_probe_sigmoid(0, chunk << 29, 1 << 29, 1, 32, out))
        assert out[0] == 0, "sigmoid_cnt3 differs from cnt / (1 + det_expf(-t)) for %d pairs, first at t bits 0x%08x cnt %d" % (
            out[0], out[1] >> 8, out[1] & 255)
        pairs += out[2]
    assert pairs == 32 << 32
    lo, hi = 0x3f800000, 0x7e800000  # mode 1: every float in [1, 2^126)
    check(lib().rto_probe_sigmoid(1, lo, hi - lo, 1, 32, out))
    assert out[0] == 0, "div_small_by_ge1 differs from the IEEE division for %d pairs, first at d bits 0x%08x cnt %d" % (
        out[0], out[1] >> 8, out[1] & 255)
    assert out[2] == 32 * (hi - lo)


def test_one_instruction_half_multiply_equals_convert_then_multiply():
    """round 6: shade_leaf_packed multiplies a basis value by an fp16 coefficient with ONE v_fma_mix_f32 (the half widened by the
    instruction, addend -0.0) instead of v_cvt_f32_f16 + v_mul_f32 -- 48 products per hit entry (rt_core.cuh:286-312).  Same
    float, checked on the device in both packed positions: every half (denormals, infinities, NaNs as NaN) against 2^20 floats
    spread over the whole range, and 16 special halves against ALL 2^32 floats."""
    import ctypes as C

    from rt_octree_amd._lib import check, lib
    out = (C.c_uint64 * 3)()
    check(lib().rto_probe_sigmoid(3, 0x00000123, 1 << 20, 1, 32, out))
    assert out[0] == 0, "mul_half differs for %d (float, half block) pairs, first at b bits 0x%08x block %d" % (out[0], out[1] >> 8, out[1] & 255)
    assert out[2] == 32 << 20
    for chunk in range(4):
        check(lib().rto_probe_sigmoid(2, chunk << 30, 1 << 30, 1, 16, out))
        assert out[0] == 0, "mul_half differs for %d (float, special half) pairs, first at b bits 0x%08x half #%d" % (
            out[0], out[1] >> 8, out[1] & 255)
        assert out[2] == 16 << 30


@pytest.mark.parametrize("fn,name", [(0, "det_logf"), (1, "det_expf"), (2, "fexp_f32")])
def test_device_math_equals_the_oracle_across_the_float_range(fn, name):
    """det_logf / det_expf / fexp_f32 stand in for the reference's `__logf` / `__expf` (DESIGN.md "Math"); device and
    oracle must agree on EVERY input, not only on those the test scenes produce: every 61st float of the whole 2^32 bit
    range (70 M values per function: all exponents, both signs, zeros, subnormals, infinities, NaNs), bit for bit."""
    import ctypes as C

    from rt_octree_amd._lib import check, lib
    import os
    stride, chunk = int(os.environ.get("RTO_MATH_SWEEP_STRIDE", "61")), 1 << 24  # 1 = all 2^32 floats (minutes)
    total = (1 << 32) // stride
    first = 0
    mism = 0
    while first < total:
        n = min(chunk, total - first)
        got = np.empty(n, np.float32)
        want = np.empty(n, np.float32)
        check(lib().rto_probe_math(fn, (first * stride) & 0xffffffff, stride, n, got.ctypes.data_as(C.c_void_p)))
        orc.lib().orc_math_sweep(fn, (first * stride) & 0xffffffff, stride, n, want.ctypes.data_as(C.c_void_p))
        g, w = got.view(np.uint32), want.view(np.uint32)
        nan = np.isnan(got) & np.isnan(want)  # any NaN payload is the same answer
        bad = np.flatnonzero((g != w) & ~nan)
        assert bad.size == 0, "%s: first mismatch at bits 0x%08x: device %r oracle %r" % (
            name, ((first + int(bad[0])) * stride) & 0xffffffff, got[bad[0]], want[bad[0]])
        mism += int(bad.size)
        first += n
    assert mism == 0
