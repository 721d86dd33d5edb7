"""Frame sharding on CPU: world_size 2 over gloo.  Each rank "renders" its poses with
the CPU oracle (the checker stands in for the GPU here: the test is about which rank renders what
with which RNG state, and about the gather), rank 0 gathers, and the result must equal the
single-process run frame for frame."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render_frames(indices):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import orc
    from rt_octree_amd import sharding, synth
    tree = synth.make_tree(depth_limit=5, basis_dim=9, seed=7)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    poses = synth.orbit_poses(7)
    W, H = 40, 32
    fx = synth.blender_focal(W)
    out = {}
    for i in indices:
        cam = orc.camera(W, H, fx, fx, poses[i][:3, :4].T.reshape(-1))
        _, rgba, _ = orc.render_frame(ht, cam, orc.default_options(spp=2), orc.rng(frame=sharding.frame_rng_jumps(i)), threads=1)
        out[i] = orc.rgba8(rgba)
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(HERE))
    from rt_octree_amd import sharding
    mine = sharding.shard_indices(7, rank, world)
    local = _render_frames(mine)
    frames = sharding.gather_frames(local, 7, rank, world, dist=dist)
    t = sharding.reduce_timings([1.0 * len(mine), 2.0 * len(mine), 3.0 * len(mine)], len(mine), dist=dist, world=world)
    if rank == 0:
        q.put((np.stack(frames), t))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_cover_all_frames_once():
    from rt_octree_amd import sharding
    for n, w in ((200, 8), (7, 2), (5, 8), (1, 1)):
        seen = sorted(i for r in range(w) for i in sharding.shard_indices(n, r, w))
        assert seen == list(range(n))
        assert all(sharding.owner_of(i, w) == r for r in range(w) for i in sharding.shard_indices(n, r, w))
    assert sharding.frame_rng_jumps(0) == 100 and sharding.frame_rng_jumps(199) == 299


def test_two_ranks_gather_equals_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    frames, timing = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _render_frames(range(7))
    for i in range(7):
        assert np.array_equal(frames[i], single[i]), "frame %d differs between 2-rank and 1-rank runs" % i
    assert timing["frames"] == 7 and abs(timing["render_ms"] - 1.0) < 1e-12 and abs(timing["fps"] - 1000.0 / 6.0) < 1e-9
