"""Round 5: the variants of the batched path that the default launch does not take must render the default's pixels.

  * the record layouts: coefficient records in the order of the two-level image's entries (TreeDev::rec_by_entry, the default
    for dense SH9 / SH16 trees) against slot-ordered records (RTO_TREE_SLOT_RECORDS=1 at upload), through every kernel --
    batched, single-frame, the counting instantiation (one-level walk + two-level lookup of the hit's entry) and the generic
    kernel after the reference arrays were rebuilt FROM the entry-ordered records;
  * a launch whose SPP leaves a hit entry too few bits for an entry of the two-level image, but enough for a leaf
    slot, walks the ONE-level image instead of dropping to the generic kernel (test hook: tuning "wide_bits")."""
import os

import numpy as np
import pytest

import orc
import rt_octree_amd as R
from helpers import assert_bits_equal, cameras, make_pair, oracle_frame
from rt_octree_amd import synth

pytestmark = pytest.mark.gpu
POSES = synth.orbit_poses(8)


def batch_frames(dt, cams, spp, jumps, tuning=(), lean=False):
    W, H = cams[0].width, cams[0].height
    ctx = R.RenderContext(W, H, frames=len(cams))
    for k, v in tuning:
        ctx.set_tuning(k, v)
    ctx.rng_seed()
    R.launch_renderer_batch(dt, cams, R.RenderOptions(spp=spp, denoise=False), ctx, rng_jumps=jumps)
    out = []
    for f in range(len(cams)):
        ctx.select_frame(f)
        out.append(ctx.download_aux())
    ctx.free()
    return np.stack(out)


def test_entry_ordered_and_slot_ordered_records_render_the_same_frames():
    tree = synth.make_tree(depth_limit=7, basis_dim=16, seed=5, shell=2.0)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    dt_e = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    os.environ["RTO_TREE_SLOT_RECORDS"] = "1"
    try:
        dt_s = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    finally:
        del os.environ["RTO_TREE_SLOT_RECORDS"]
    assert dt_e.wide_nodes > 0 and dt_s.wide_nodes == dt_e.wide_nodes
    assert dt_e.device_bytes > dt_s.device_bytes  # (a first-level leaf of a pair owns 8 entries: more records than slots)
    #  when the device refuses the larger, entry-ordered copy the upload retries with slot-ordered records instead
    # of giving the aligned records up altogether (test hook: RTO_TEST_FAIL_ENTRY_RECORDS makes that first allocation "fail")
    os.environ["RTO_TEST_FAIL_ENTRY_RECORDS"] = "1"
    try:
        dt_r = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    finally:
        del os.environ["RTO_TEST_FAIL_ENTRY_RECORDS"]
    assert dt_r.device_bytes == dt_s.device_bytes and dt_r.wide_nodes == dt_e.wide_nodes
    W, H, spp = 160, 120, 6
    cams = [cameras(W, H, p)[1] for p in POSES[:4]]
    jumps = [100 + i for i in range(4)]
    a_e, a_s = batch_frames(dt_e, cams, spp, jumps), batch_frames(dt_s, cams, spp, jumps)
    assert_bits_equal(a_e, a_s, "batched frames, entry- vs slot-ordered records")
    assert_bits_equal(batch_frames(dt_r, cams, spp, jumps), a_e, "batched frames after the retry with slot-ordered records")
    dt_r.free()
    want = oracle_frame(ht, cameras(W, H, POSES[2])[0], spp, frame=102)
    assert_bits_equal(a_e[2], want[0], "entry-ordered records vs the oracle")
    for dt in (dt_e, dt_s):
        ctx = R.RenderContext(W, H)
        for kernel, stats in ((R.KERNEL_FAST, False), (R.KERNEL_FAST, True), (R.KERNEL_GENERIC, False)):
            ctx.rng_seed()
            ctx.rng_advance(102 << 32)
            ctx.set_kernel(kernel)
            ctx.enable_stats(stats)
            R.launch_renderer(dt, cams[2], R.RenderOptions(spp=spp, denoise=False), ctx)
            assert_bits_equal(ctx.download_aux(), want[0], "kernel %d stats %d" % (kernel, stats))
        ctx.enable_stats(False)
        ctx.free()
    dt_e.free()
    dt_s.free()


def test_two_level_image_over_budget_walks_the_one_level_image():
    """slot-ordered records (compact records / RGBA trees have them): with a hit entry 'too small' for the two-level image's
    entries the launch takes the WIDE = false instantiations -- same frames, no generic fallback (the generic kernel would
    need the reference arrays back: device_bytes stays put)"""
    tree = synth.make_tree(depth_limit=7, basis_dim=9, seed=3, shell=2.0)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format, compact_records=True)
    assert dt.wide_nodes > 0
    W, H, spp = 160, 120, 6
    cams = [cameras(W, H, p)[1] for p in POSES[:3]]
    jumps = [5, 6, 7]
    base = batch_frames(dt, cams, spp, jumps)
    got = batch_frames(dt, cams, spp, jumps, tuning=(("wide_bits", 6),))  # 2^6 entries: nothing fits
    assert_bits_equal(got, base, "one-level walk forced by the budget hook")
    one = R.RenderContext(W, H)
    one.set_tuning("wide_bits", 6)
    one.rng_seed()
    one.rng_advance(6 << 32)
    R.launch_renderer(dt, cams[1], R.RenderOptions(spp=spp, denoise=False), one)
    assert_bits_equal(one.download_aux(), base[1], "single-frame kernel on the one-level image")
    # a tree whose records follow the entries cannot fall back (its records are indexed by them): generic kernel, same frames
    dt_e = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    got_e = batch_frames(dt_e, cams, spp, jumps, tuning=(("wide_bits", 6),))
    assert_bits_equal(got_e, base, "entry-ordered records over budget: the generic kernel, same frames")
    one.free()
    dt.free()
    dt_e.free()


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5, 6, 7, 8, -7, -8, -10, -11])
def test_every_tree_depth_through_both_stack_forms(depth):
    """late round 5: the restart of a march step by coordinate-difference thresholds (register-stack form: at most two pairs
    of levels below the top grid) and the LDS-stack form beside it.  The grid spans min(depth - 1, 6) levels (none below
    depth 3), so these depths cover: no grid at all, a grid with one half-filled pair below it, one pair, two pairs (the
    benchmark's shape), three (LDS rows) -- batched and single-frame kernels against the oracle, bit for bit.  (Negative:
    a deep narrow chain tree of that many levels, test_render_parity._chain_tree -- a full tree of depth 10 takes minutes to make.)"""
    if depth < 0:
        from test_render_parity import _chain_tree
        tree = _chain_tree(-depth, seed=-depth, basis=9)
    else:
        tree = synth.make_tree(depth_limit=depth, basis_dim=9, seed=40 + depth, shell=2.0)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    W, H, spp = 96, 72, 4
    cams = [cameras(W, H, p)[1] for p in POSES[:3]]
    jumps = [100 + i for i in range(3)]
    got = batch_frames(dt, cams, spp, jumps)
    for f in (0, 2):
        want = oracle_frame(ht, cameras(W, H, POSES[f])[0], spp, frame=100 + f)
        assert_bits_equal(got[f], want[0], "batched kernel, depth %d, frame %d" % (depth, f))
    ctx = R.RenderContext(W, H)
    ctx.set_kernel(R.KERNEL_FAST)
    ctx.rng_seed()
    ctx.rng_advance(102 << 32)
    R.launch_renderer(dt, cams[2], R.RenderOptions(spp=spp, denoise=False), ctx)
    assert_bits_equal(ctx.download_aux(), got[2], "single-frame kernel, depth %d" % depth)
    ctx.free()
    dt.free()


@pytest.mark.parametrize("seed", range(24))
def test_two_pairs_below_the_grid_random_views(seed):
    """the register-stack restart's one non-trivial move -- leaving the SECOND pair of levels for the first pair's node without
    going back to the grid -- needs trees with two pairs below the grid (depth 9-10), which the fuzz scenes (depth <= 7) do not
    have: deep narrow trees, random cameras outside, inside and grazing the box, batched and single-frame kernels against the oracle"""
    from test_render_parity import _chain_tree
    rs = np.random.RandomState(7000 + seed)
    depth = int(rs.choice([9, 10]))
    tree = _chain_tree(depth, seed=int(rs.randint(0, 64)), basis=int(rs.choice([4, 9])))
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    assert dt.max_depth == depth
    W, H = int(rs.randint(24, 90)), int(rs.randint(24, 70))
    spp = int(rs.choice([1, 4, 6, 8]))
    ocams, cams = [], []
    for _ in range(3):
        r = rs.uniform(0.2, 4.0)
        pos = rs.randn(3)
        pos = pos / np.linalg.norm(pos) * r
        oc, c = cameras(W, H, synth.look_at_c2w(pos, rs.uniform(-0.4, 0.4, 3)))
        ocams.append(oc)
        cams.append(c)
    jumps = [5 + i for i in range(3)]
    got = batch_frames(dt, cams, spp, jumps)
    for f in range(3):
        want = oracle_frame(ht, ocams[f], spp, frame=jumps[f])
        assert_bits_equal(got[f], want[0], "batched kernel, depth %d, seed %d, frame %d" % (depth, seed, f))
    ctx = R.RenderContext(W, H)
    ctx.set_kernel(R.KERNEL_FAST)
    ctx.rng_seed()
    ctx.rng_advance(jumps[1] << 32)
    R.launch_renderer(dt, cams[1], R.RenderOptions(spp=spp, denoise=False), ctx)
    assert_bits_equal(ctx.download_aux(), got[1], "single-frame kernel, depth %d, seed %d" % (depth, seed))
    ctx.free()
    dt.free()


def test_top_grid_levels_never_change_pixels():
    """round 6 A/B hook RTO_TOP_LEVELS (levels the top grid of the traversal images covers, default 6): derived data, so the
    frames of the batched and the single-frame kernels are the default's for every value"""
    tree = synth.make_tree(depth_limit=9, basis_dim=9, seed=21, shell=2.0)
    W, H, spp = 200, 152, 6
    cams = [cameras(W, H, p)[1] for p in POSES[:3]]
    jumps = [11, 12, 13]
    base = None
    for g in (None, 4, 5, 7, 8):
        if g is not None:
            os.environ["RTO_TOP_LEVELS"] = str(g)
        try:
            dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
        finally:
            os.environ.pop("RTO_TOP_LEVELS", None)
        got = batch_frames(dt, cams, spp, jumps)
        ctx = R.RenderContext(W, H)
        ctx.rng_seed()
        ctx.rng_advance(jumps[1] << 32)
        R.launch_renderer(dt, cams[1], R.RenderOptions(spp=spp, denoise=False), ctx)
        one = ctx.download_aux()
        ctx.free()
        dt.free()
        if base is None:
            base = got
            assert np.any(base[:, 3] > 0)
        assert_bits_equal(got, base, "batched frames with a top grid of %s levels" % g)
        assert_bits_equal(one, base[1], "single-frame kernel with a top grid of %s levels" % g)
