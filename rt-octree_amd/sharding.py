"""Frame sharding across the GPUs of one node.

The render path has no exchange step: pixels are independent within a frame and frames are
independent across poses (the only cross-frame state is the RNG, which is jump-ahead addressable:
the reference's timed frame i uses `pcg32(20230418)` advanced (100 + i) times by 2^32,
main_headless.cpp:469-506).  So: one process per GPU, a replica of the tree per GPU, pose i goes to
rank i mod N, and every image is bit-identical to the single-GPU run whatever N is.  The only
collective is the final gather of the RGBA8 frames to rank 0 (RCCL over xGMI on GPUs; gloo in the
CPU tests) plus the reduction of the timing scalars.
"""
import numpy as np

WARM_FRAMES_REF = 100  # main_headless.cpp:469-479


def shard_indices(n_frames, rank, world):
    """global frame indices rendered by `rank`: i with i mod world == rank, ascending"""
    return list(range(rank, n_frames, world))


def frame_rng_jumps(frame_index, warm_frames=WARM_FRAMES_REF):
    """number of 2^32 jumps that bring RenderContext.rng to the state the reference uses for the
    timed frame `frame_index` (each warm-up and each rendered frame advances once)"""
    return warm_frames + frame_index


def owner_of(frame_index, world):
    return frame_index % world


def gather_frames(local_frames, n_frames, rank, world, dist=None, device=None):
    """local_frames: {global_index: uint8 array [H,W,4]} of this rank.  Returns the list of all
    n_frames arrays on rank 0 (None elsewhere).  One padded all_gather of a [max_local,H,W,4] uint8
    tensor: 2.56 MB per 800x800 frame, far below one xGMI link-second even for 200 frames."""
    if world == 1:
        return [local_frames[i] for i in range(n_frames)]
    import torch
    if dist is None:
        import torch.distributed as dist
    mine = shard_indices(n_frames, rank, world)
    max_local = (n_frames + world - 1) // world
    if n_frames < world:
        # fewer frames than ranks (--max_imgs 5 over 8 GPUs): some ranks own no frame and so know no frame shape -- every rank
        # learns it from the ones that do (the condition is the same on all ranks, so they all take this extra collective)
        shp = torch.tensor(list(next(iter(local_frames.values())).shape) if local_frames else [0, 0, 0], dtype=torch.int64)
        if device is not None:
            shp = shp.to(device)
        dist.all_reduce(shp, op=dist.ReduceOp.MAX)
        H, W, C = (int(v) for v in shp.cpu())
    else:
        H, W, C = next(iter(local_frames.values())).shape
    buf = torch.zeros((max_local, H, W, C), dtype=torch.uint8)
    for k, i in enumerate(mine):
        buf[k] = torch.from_numpy(np.ascontiguousarray(local_frames[i]))
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    if rank != 0:
        return None
    frames = [None] * n_frames
    for r in range(world):
        arr = out[r].cpu().numpy()
        for k, i in enumerate(shard_indices(n_frames, r, world)):
            frames[i] = arr[k]
    return frames


def reduce_timings(ms_sum, frames, dist=None, device=None, world=1):
    """sum of per-rank (ms_sum[3], frames) -> global mean ms per stage and the reference FPS formula"""
    import torch
    t = torch.tensor(list(ms_sum) + [float(frames)], dtype=torch.float64)
    if world > 1:
        if dist is None:
            import torch.distributed as dist
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t)
        t = t.cpu()
    n = max(float(t[3]), 1.0)
    mean = [float(t[i]) / n for i in range(3)]
    total = sum(mean)
    return {"render_ms": mean[0], "torch_ms": mean[1], "filter_ms": mean[2], "fps": 1000.0 / total if total > 0 else 0.0,
            "frames": int(t[3])}
