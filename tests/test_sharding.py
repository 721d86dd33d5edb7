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


def _empty_rank_worker(rank, world, port, q, n_frames):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(HERE))
    from rt_octree_amd import sharding
    # (synthetic frames: what is under test is the collective, and 8 processes must stay light on an 8-core box)
    local = {i: np.full((6, 5, 4), 10 * i + 1, np.uint8) for i in sharding.shard_indices(n_frames, rank, world)}
    frames = sharding.gather_frames(local, n_frames, rank, world, dist=dist)
    if rank == 0:
        q.put(np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_with_ranks_that_own_no_frame():
    """`--max_imgs 5` over 8 ranks -- ranks 5..7 own no frame (and know no frame shape); the gather must neither
    raise there nor hang the others"""
    world, n_frames = 8, 5
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, world, port, q, n_frames)) for r in range(world)]
    for p in procs:
        p.start()
    frames = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert frames.shape == (n_frames, 6, 5, 4)
    for i in range(n_frames):
        assert np.all(frames[i] == 10 * i + 1)


# ------------------------------------------------------------------ bench.py's frame schedule
def _bench():
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib
    return importlib.import_module("bench")


def test_bench_schedule_partitions_frames_over_ranks():
    """pose i -> rank i mod N (one scene); with K scenes both mappings render every (scene, pose) of a full
    cycle exactly once, and a launch group never mixes scenes."""
    b = _bench()
    for world in (1, 2, 4, 8):
        seen = sorted(b.pose_schedule(s, r, world, 200, 1, "pose") for r in range(world) for s in range(200 // world))
        assert seen == [(0, p) for p in range(200)]
        assert all(b.pose_schedule(s, r, world, 200, 1, "pose")[1] % world == r for r in range(world) for s in range(25))
    K = 8
    for world in (1, 2, 4, 8):
        for m in ("pose", "scene"):
            per_rank = K * 200 // world
            seen = sorted(b.pose_schedule(s, r, world, 200, K, m) for r in range(world) for s in range(per_rank))
            assert seen == [(sc, p) for sc in range(K) for p in range(200)], (world, m)
            for r in range(world):
                assert {sc for sc, _ in (b.pose_schedule(s, r, world, 200, K, m) for s in range(per_rank))} == set(b.scenes_of_rank(r, world, K, m))
                for sc, idx in b.plan_groups(per_rank, 32, r, world, 200, K, m):
                    assert 1 <= len(idx) <= 32 and len(set(idx)) == len(idx)
    # N ranks render an orbit N times as dense: a 100-frame launch group never holds a pose twice, and a rank's consecutive
    # poses are as far apart as the 1-GPU run's (1/200 of the orbit)
    for world in (1, 2, 4, 8):
        n = b.n_poses_for(world)
        assert n == 200 * world
        for r in range(world):
            for sc, idx in b.plan_groups(300, 100, r, world, n, 1, "pose"):
                assert len(set(idx)) == len(idx) == 100 and all(p % world == r for p in idx)
                steps = {(idx[k + 1] - idx[k]) % n for k in range(len(idx) - 1)}
                assert steps == {world}
    # more ranks than scenes: every rank still has work
    assert b.scenes_of_rank(5, 8, 3, "scene") == [2]


def test_bench_plan_two_ranks_over_gloo(tmp_path):
    """`torchrun --nproc-per-node 2 bench.py --plan-only`: the driver's launch line on a CPU box, up to (not
    including) the GPU calls -- RANK / WORLD_SIZE parsing, rendezvous on 127.0.0.1, the exchanged plans."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--batch", "32", "--groups-per-step", "1",
           "--warmup", "0", "--scenes", "2", "--plan-only"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    doc = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert doc["world"] == 2 and doc["scenes"] == 2
    for m in ("pose", "scene"):
        assert len(doc["plans"][m]) == 2
        frames = [[(sc, p) for sc, idx in rank_plan for p in idx] for rank_plan in doc["plans"][m]]
        assert all(len(f) == 96 for f in frames)
        assert not set(frames[0]) & set(frames[1])  # no frame is rendered twice
    # pose map: rank r renders the poses with p mod 2 == r of scene 0 (192 global frames < 200 poses)
    assert all(p % 2 == r for r in range(2) for _, idx in doc["plans"]["pose"][r] for p in idx)
    # scene map: rank r renders scene r
    assert all(sc == r for r in range(2) for sc, _ in doc["plans"]["scene"][r])


def _nccl_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import rt_octree_amd as R
    from rt_octree_amd import sharding, synth
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    tree = synth.make_tree(depth_limit=5, basis_dim=9, seed=7)
    dt = R.N3Tree.from_arrays(tree.child, tree.data, tree.scale, tree.offset, tree.data_format, device=rank)
    poses = synth.orbit_poses(7)
    W, H = 40, 32
    fx = synth.blender_focal(W)
    ctx = R.RenderContext(W, H, device=rank)
    local = {}
    for i in sharding.shard_indices(7, rank, world):
        cam = R.Camera(W, H, fx, fx)
        cam.set_c2w(poses[i])
        ctx.rng_seed()
        ctx.rng_advance(sharding.frame_rng_jumps(i) << 32)
        R.launch_renderer(dt, cam, R.RenderOptions(spp=2, denoise=False), ctx)
        local[i] = ctx.download_rgba8()
    frames = sharding.gather_frames(local, 7, rank, world, dist=dist, device=dev)  # RCCL all_gather of device tensors
    if rank == 0:
        q.put(np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_gpus_gather_over_rccl_equals_single_gpu():
    """The final gather over RCCL with device tensors, one process per GPU.  Needs two GPUs: skipped on the
    1-GPU test box, exercised by the first multi-GPU run."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    frames = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _render_frames(range(7))  # the CPU oracle, bit-exact with the HIP kernels
    for i in range(7):
        assert np.array_equal(frames[i], single[i]), "frame %d differs between the 2-GPU run and the oracle" % i
