"""Seeded synthetic inputs in the reference's own file formats (no dataset
ships with the reference and none exists on the GPU box).

  * a "lego-like" PlenOctree: N = 2, refined only around a thin shell of an analytic SDF scene
    (plates, boxes, studs, wheels, a boom), sigma > 0 only in shell leaves, SH-9 / SH-16 colours;
    written with the key schema N3Tree::load_npz reads (n3tree.cpp:228-362);
  * blender-schema camera paths (main_headless.cpp:255-272), T&T-style pose directories
    (:273-297) and the reference's options/opt.json.

Pure numpy; used by bench.py, the tests and tools/.  Not part of the render path.
"""
import json
import os

import numpy as np

CAMERA_ANGLE_X = 0.6911112070083618  # NeRF-synthetic; fx = 0.5*W/tan(0.5*angle) = 1111.11 @ 800


# ------------------------------------------------------------------ scene SDF (world units)
def _sd_box(p, c, h):
    q = np.abs(p - np.asarray(c, np.float32)) - np.asarray(h, np.float32)
    return np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(np.max(q, axis=-1), 0)


def _sd_cyl(p, c, r, hh, axis):
    d = p - np.asarray(c, np.float32)
    other = [i for i in range(3) if i != axis]
    dr = np.sqrt(d[..., other[0]] ** 2 + d[..., other[1]] ** 2) - r
    da = np.abs(d[..., axis]) - hh
    return np.minimum(np.maximum(dr, da), 0) + np.sqrt(np.maximum(dr, 0) ** 2 + np.maximum(da, 0) ** 2)


def scene_sdf(p):
    """Union of primitives, roughly a toy bulldozer inside [-1.1, 1.1]^3 (z up)."""
    p = p.astype(np.float32, copy=False)
    d = _sd_box(p, (0, 0, -0.52), (1.05, 0.65, 0.05))                       # base plate
    d = np.minimum(d, _sd_box(p, (-0.05, 0, -0.22), (0.62, 0.36, 0.22)))    # body
    d = np.minimum(d, _sd_box(p, (-0.25, 0, 0.2), (0.3, 0.3, 0.2)))         # cab
    d = np.minimum(d, _sd_box(p, (0.78, 0, -0.28), (0.06, 0.55, 0.2)))      # blade
    for sy in (-0.5, 0.5):                                                    # tracks + wheels
        d = np.minimum(d, _sd_box(p, (-0.05, sy, -0.36), (0.7, 0.09, 0.1)))
        for wx in (-0.6, -0.2, 0.2, 0.6):
            d = np.minimum(d, _sd_cyl(p, (wx, sy, -0.36), 0.13, 0.11, 1))
    # boom: a thin box rotated about y
    ang = np.float32(0.6)
    ca, sa = np.cos(ang), np.sin(ang)
    q = p - np.asarray((0.25, 0.0, 0.25), np.float32)
    qr = np.stack([ca * q[..., 0] + sa * q[..., 2], q[..., 1], -sa * q[..., 0] + ca * q[..., 2]], -1)
    d = np.minimum(d, _sd_box(qr, (0, 0, 0), (0.5, 0.05, 0.05)))
    # studs on the plate and on the cab roof
    for sx in np.arange(-0.9, 0.91, 0.3):
        for sy in (-0.3, 0.0, 0.3):
            d = np.minimum(d, _sd_cyl(p, (sx, sy, -0.44), 0.06, 0.035, 2))
    for sx in (-0.4, -0.1):
        for sy in (-0.15, 0.15):
            d = np.minimum(d, _sd_cyl(p, (sx, sy, 0.43), 0.06, 0.035, 2))
    return d


def scene_variant(k):
    """SDF of synthetic scene k (config C3's "8 scenes"): scene 0 is scene_sdf itself, scene k > 0 the same
    model turned about the z axis, uniformly scaled and with a different part removed -- other topology,
    other node counts, other ray statistics."""
    if k == 0:
        return scene_sdf
    ang = np.float32(0.7 * k)
    ca, sa = np.cos(ang), np.sin(ang)
    scl = np.float32(0.78 + 0.045 * k)
    cut = np.asarray([(0.8, 0.0, -0.3), (-0.3, 0.0, 0.3), (0.0, 0.5, -0.35), (0.0, -0.5, -0.35)][k % 4], np.float32)

    def sdf(p):
        p = p.astype(np.float32, copy=False)
        q = np.stack([ca * p[..., 0] + sa * p[..., 1], -sa * p[..., 0] + ca * p[..., 1], p[..., 2]], -1) / scl
        d = scene_sdf(q)
        hole = np.linalg.norm(q - cut, axis=-1) - np.float32(0.28)  # carve a ball out of the model
        return np.maximum(d, -hole) * scl
    return sdf


# ------------------------------------------------------------------ tree
class SynthTree:
    """Host arrays of a synthetic PlenOctree, in the reference's layout."""

    def __init__(self, child, data, scale, offset, data_format, depth_limit, stats):
        self.child, self.data = child, data
        self.scale = np.asarray(scale, np.float32)
        self.offset = np.asarray(offset, np.float32)
        self.data_format = data_format
        self.depth_limit = depth_limit
        self.stats = stats

    @property
    def capacity(self):
        return self.child.shape[0]

    @property
    def data_dim(self):
        return self.data.shape[-1]

    def save_npz(self, path, compressed=False):
        """Key schema of svox / N3Tree::load_npz (n3tree.cpp:228-362)."""
        kw = dict(data_dim=np.int64(self.data_dim), data_format=np.array(self.data_format),
                  invradius3=self.scale.astype(np.float32), offset=self.offset.astype(np.float32),
                  child=self.child, data=self.data)
        (np.savez_compressed if compressed else np.savez)(path, **kw)
        return path

    def save_quant_npz(self, path, n_retain=1, compressed=False, quantiser="luminance"):
        """The quantised schema of octree/compression.py (decoded by n3tree.cpp:279-340): basis
        functions 0..n_retain-1 stay fp16 (`data_retained`), every other basis function gets a
        65536-entry RGB codebook (`quant_colors`) + a uint16 index per leaf slot (`quant_map`).
        quantiser "luminance": a cheap stand-in -- slots ordered by luminance of the coefficient triple and
        cut into 65536 equal-count cells, codebook entry = cell mean.  "median_cut" (round 6): Heckbert's
        median cut over the triples of the leaves with sigma > 0, as renderer/scripts/compress_octree.py:68-119
        applies svox's `quantize_median_cut` (that C extension is not in the reference tree; this is the published
        algorithm: 16 rounds of splitting every box at the median of its widest axis, palette = box means, id =
        the box's position in the split tree; slots with sigma = 0 get id 0 and sigma 0).
        Returns the decoded dense data (what N3Tree::load_npz would expand to)."""
        if quantiser == "median_cut":
            return self._save_quant_median_cut(path, n_retain, compressed)
        assert quantiser == "luminance"
        cap, D = self.capacity, self.data_dim
        nb = (D - 1) // 3
        assert self.data_format.startswith("SH") and 0 <= n_retain <= nb
        n_child = cap * 8
        flat = self.data.reshape(n_child, D)
        nq = nb - n_retain
        retained = np.zeros((n_retain, n_child, 3), np.float16)
        for k in range(n_retain):
            for c in range(3):
                retained[k, :, c] = flat[:, c * nb + k]
        qcolors = np.zeros((nq, 65536, 3), np.float16)
        qmap = np.zeros((nq, n_child), np.uint16)
        for j in range(nq):
            k = j + n_retain
            tri = np.stack([flat[:, c * nb + k] for c in range(3)], 1).astype(np.float32)
            order = np.argsort(tri @ np.array([0.299, 0.587, 0.114], np.float32), kind="stable")
            cell = (np.arange(n_child, dtype=np.int64) * 65536 // n_child).astype(np.int64)
            ids = np.empty(n_child, np.int64)
            ids[order] = cell
            cnt = np.maximum(np.bincount(ids, minlength=65536), 1)
            for c in range(3):
                qcolors[j, :, c] = (np.bincount(ids, weights=tri[:, c], minlength=65536) / cnt).astype(np.float16)
            qmap[j] = ids.astype(np.uint16)
        sigma = np.ascontiguousarray(flat[:, D - 1]).reshape(cap, 2, 2, 2)
        kw = dict(data_dim=np.int64(D), data_format=np.array(self.data_format),
                  invradius3=self.scale.astype(np.float32), offset=self.offset.astype(np.float32),
                  child=self.child, quant_colors=qcolors, quant_map=qmap.reshape(nq, cap, 2, 2, 2), sigma=sigma)
        if n_retain:
            kw["data_retained"] = retained.reshape(n_retain, cap, 2, 2, 2, 3)
        (np.savez_compressed if compressed else np.savez)(path, **kw)
        dec = np.zeros((n_child, D), np.float16)
        for k in range(n_retain):
            for c in range(3):
                dec[:, c * nb + k] = retained[k, :, c]
        for j in range(nq):
            col = qcolors[j][qmap[j]]
            for c in range(3):
                dec[:, c * nb + j + n_retain] = col[:, c]
        dec[:, D - 1] = flat[:, D - 1]
        return dec.reshape(self.data.shape)


def median_cut_ids(tri, bits=16, device=None):
    """Heckbert median cut of the rows of tri [m, 3] (float32) into 2^bits boxes: `bits` rounds, every box split at the median
    of its widest axis.  -> (palette [2^bits, 3] float32 = box means, ids [m] int64 = box of each row, in split-tree order).
    torch, on the GPU when there is one (a round is one 64-bit sort of m keys)."""
    import torch
    dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
    x = torch.as_tensor(np.ascontiguousarray(tri, np.float32), device=dev)
    m = x.shape[0]
    box = torch.zeros(m, dtype=torch.int64, device=dev)
    for lvl in range(bits):
        nb = 1 << lvl
        lo = torch.full((nb, 3), float("inf"), device=dev).scatter_reduce(0, box[:, None].expand(-1, 3), x, "amin")
        hi = torch.full((nb, 3), float("-inf"), device=dev).scatter_reduce(0, box[:, None].expand(-1, 3), x, "amax")
        axis = torch.argmax(hi - lo, dim=1)                       # widest axis of every box (empty boxes: any)
        val = x.gather(1, axis[box][:, None])[:, 0]
        # order the rows by (box, value along the box's axis): one sort of a 64-bit key
        u = val.view(torch.int32).to(torch.int64)
        u = torch.where(u < 0, ~u & 0x7fffffff, u | 0x80000000)  # float bits -> unsigned keys of the same order
        order = torch.argsort((box << 32) | u)
        sb = box[order]
        cnt = torch.bincount(sb, minlength=nb)
        start = torch.cumsum(cnt, 0) - cnt
        rank = torch.arange(m, device=dev) - start[sb]
        upper = rank >= (cnt[sb] + 1) // 2                        # the median stays in the lower half
        nbx = torch.empty_like(box)
        nbx[order] = sb * 2 + upper.to(torch.int64)
        box = nbx
    nb = 1 << bits
    cnt = torch.bincount(box, minlength=nb).clamp(min=1).to(torch.float32)
    pal = torch.zeros((nb, 3), device=dev).index_add_(0, box, x) / cnt[:, None]
    return pal.cpu().numpy(), box.cpu().numpy()


def _save_quant_median_cut(self, path, n_retain, compressed):
    cap, D = self.capacity, self.data_dim
    nb = (D - 1) // 3
    assert self.data_format.startswith("SH") and 0 <= n_retain <= nb
    n_child = cap * 8
    flat = self.data.reshape(n_child, D)
    sig = flat[:, D - 1].astype(np.float32)
    snz = sig > 0.0  # compress_octree.py:72-73 (its --sigma_thresh; leaves below it lose their density)
    nq = nb - n_retain
    retained = np.zeros((n_retain, n_child, 3), np.float16)
    for k in range(n_retain):
        for c in range(3):
            retained[k, snz, c] = flat[snz, c * nb + k]
    qcolors = np.zeros((nq, 65536, 3), np.float16)
    qmap = np.zeros((nq, n_child), np.uint16)
    for j in range(nq):
        k = j + n_retain
        tri = np.stack([flat[snz, c * nb + k] for c in range(3)], 1).astype(np.float32)
        pal, ids = median_cut_ids(tri)
        qcolors[j] = pal.astype(np.float16)
        qmap[j, snz] = ids.astype(np.uint16)
    sigma = np.where(snz, flat[:, D - 1], np.float16(0)).astype(np.float16).reshape(cap, 2, 2, 2)
    kw = dict(data_dim=np.int64(D), data_format=np.array(self.data_format),
              invradius3=self.scale.astype(np.float32), offset=self.offset.astype(np.float32),
              child=self.child, quant_colors=qcolors, quant_map=qmap.reshape(nq, cap, 2, 2, 2), sigma=sigma)
    if n_retain:
        kw["data_retained"] = retained.reshape(n_retain, cap, 2, 2, 2, 3)
    (np.savez_compressed if compressed else np.savez)(path, **kw)
    dec = np.zeros((n_child, D), np.float16)
    for k in range(n_retain):
        for c in range(3):
            dec[:, c * nb + k] = retained[k, :, c]
    for j in range(nq):
        col = qcolors[j][qmap[j]]
        for c in range(3):
            dec[:, c * nb + j + n_retain] = col[:, c]
    dec[:, D - 1] = sigma.reshape(-1)
    return dec.reshape(self.data.shape)


SynthTree._save_quant_median_cut = _save_quant_median_cut


def make_tree(depth_limit=6, basis_dim=9, seed=20230418, shell=1.25, radius=1.5, sdf=scene_sdf,
              max_nodes=None):
    """Refine cells whose centre lies within (half-diagonal + shell*finest_cell) of the surface.

    depth_limit D: nodes live at levels 0..D-1, finest cells are 2^-D (tree units) across.
    Node order is breadth-first, children appended behind their parents, child[] holds the
    relative node offset (0 = leaf) exactly as svox writes it.
    """
    rng = np.random.default_rng(seed)
    offset = np.full(3, 0.5, np.float32)
    scale = np.full(3, 1.0 / (2.0 * radius), np.float32)  # invradius3
    data_dim = 3 * basis_dim + 1
    fine = 2.0 ** -depth_limit
    tau = shell * fine  # shell half-thickness, tree units

    def sdf_tree(pt):  # tree units in, tree units out (uniform scale)
        return sdf((pt - offset) / scale) * scale[0]

    corner = np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], np.int64)  # x major
    child_levels, occ_levels, ctr_levels = [], [], []
    coords = np.zeros((1, 3), np.int64)  # integer cell coords of the nodes at this level
    n_nodes_before = 0
    for lvl in range(depth_limit):
        n = coords.shape[0]
        cell = 2.0 ** -(lvl + 1)
        sub = (coords[:, None, :] * 2 + corner[None, :, :])                     # [n,8,3]
        centers = ((sub.astype(np.float32) + 0.5) * np.float32(cell)).reshape(-1, 3)
        dist = np.abs(sdf_tree(centers)).reshape(n, 8)
        half_diag = np.float32(cell * 0.8660254)
        refine = (dist <= half_diag + tau) if lvl + 1 < depth_limit else np.zeros((n, 8), bool)
        if max_nodes is not None and n_nodes_before + n + int(refine.sum()) > max_nodes:
            refine[:] = False
        occupied = (~refine) & (dist <= tau)
        child = np.zeros((n, 8), np.int32)
        n_new = int(refine.sum())
        if n_new:
            first_child = n_nodes_before + n                       # BFS: next level starts here
            new_ids = first_child + np.arange(n_new, dtype=np.int64)
            node_ids = n_nodes_before + np.arange(n, dtype=np.int64)
            rel = np.zeros((n, 8), np.int64)
            rel[refine] = new_ids
            rel -= np.where(refine, node_ids[:, None], 0)
            child = rel.astype(np.int32)
        child_levels.append(child)
        occ_levels.append(occupied)
        ctr_levels.append(centers.reshape(n, 8, 3))
        n_nodes_before += n
        coords = sub[refine]
        if n_new == 0:
            break

    capacity = n_nodes_before
    child = np.concatenate(child_levels, 0).reshape(capacity, 2, 2, 2)
    occupied = np.concatenate(occ_levels, 0).reshape(-1)
    centers = np.concatenate(ctr_levels, 0).reshape(-1, 3)
    data = np.zeros((capacity * 8, data_dim), np.float16)
    occ_idx = np.flatnonzero(occupied)
    B = basis_dim
    # SH band of each coefficient index -> amplitude decay
    band = np.array([0] + [1] * 3 + [2] * 5 + [3] * 7 + [4] * 9, np.float32)[:B]
    amp = np.where(band == 0, 0.0, 0.6 / (1.0 + band) ** 1.5).astype(np.float32)
    CH = 1 << 20
    for s in range(0, occ_idx.size, CH):
        idx = occ_idx[s:s + CH]
        c = centers[idx]
        m = idx.size
        # Appearance: smooth fields of position (what a trained PlenOctree looks like at the scale of
        # a pixel) plus a little per-leaf noise -- a denoiser can only trade variance for resolution
        # where neighbouring pixels see related colours.  The random draws keep their order and sizes,
        # so sigma below (everything the traversal reads) does not depend on these choices.
        noise = rng.standard_normal((m, 3, B), dtype=np.float32)
        kk = np.arange(B, dtype=np.float32)
        coef = np.empty((m, 3, B), np.float32)
        for ch in range(3):
            fxk, fyk, fzk = 3.0 + 1.7 * ((kk + ch) % 4), 2.5 + 1.3 * ((2 * kk + ch) % 5), 3.5 + 1.1 * ((3 * kk + 2 * ch) % 3)
            field = np.sin(c[:, 0:1] * fxk[None] + 0.9 * kk[None] + ch) * np.cos(c[:, 1:2] * fyk[None] - 0.4 * kk[None]) \
                + 0.5 * np.sin(c[:, 2:3] * fzk[None] + 1.7 * ch)
            coef[:, ch, :] = (field + 0.1 * noise[:, ch, :]) * amp[None, :]
        # smooth DC colour field + a little per-leaf noise (pre-sigmoid, DC basis = 0.282)
        for ch, (fx, fy, fz, ph) in enumerate(((9.0, 5.0, 7.0, 0.0), (6.0, 11.0, 4.0, 1.3), (5.0, 8.0, 10.0, 2.1))):
            dc = 2.2 * np.sin(fx * c[:, 0] + ph) * np.cos(fy * c[:, 1] - ph) + 1.5 * np.sin(fz * c[:, 2] + 2 * ph)
            coef[:, ch, 0] = (dc + 0.03 * rng.standard_normal(m, dtype=np.float32)) / 0.28209479
        sigma = np.exp(rng.uniform(np.log(5.0), np.log(300.0), m)).astype(np.float32)
        rec = np.concatenate([coef.reshape(m, 3 * B), sigma[:, None]], 1)
        data[idx] = rec.astype(np.float16)
    data = data.reshape(capacity, 2, 2, 2, data_dim)
    stats = {"capacity": int(capacity), "leaf_slots": int((child == 0).sum()),
             "occupied_leaves": int(occ_idx.size), "depth_limit": int(depth_limit),
             "levels": [int(c.shape[0]) for c in child_levels]}
    return SynthTree(child, data, scale, offset, "SH%d" % basis_dim, depth_limit, stats)


def shuffle_nodes(tree, seed=1):
    """The same octree with its nodes stored in a random order (the root stays node 0), child[] offsets
    recomputed.  svox-refined trees carry no ordering guarantee (refine appends the children of whatever
    leaves were selected); make_tree's breadth-first order is the friendliest possible one.  Every query
    returns the same leaf values, so every image is unchanged."""
    cap = tree.capacity
    rng = np.random.default_rng(seed)
    new_of_old = np.concatenate([[0], 1 + rng.permutation(cap - 1)]).astype(np.int64)
    child = tree.child.reshape(cap, 8).astype(np.int64)
    node = np.arange(cap, dtype=np.int64)[:, None]
    tgt_old = node + child                                   # old index of the child node (where child != 0)
    rel_new = np.where(child != 0, new_of_old[np.where(child != 0, tgt_old, 0)] - new_of_old[node], 0)
    child_new = np.empty((cap, 8), np.int32)
    child_new[new_of_old] = rel_new.astype(np.int32)
    data_new = np.empty_like(tree.data)
    data_new[new_of_old] = tree.data
    stats = dict(tree.stats, shuffled_seed=int(seed))
    return SynthTree(child_new.reshape(cap, 2, 2, 2), data_new, tree.scale, tree.offset, tree.data_format,
                     tree.depth_limit, stats)


# ------------------------------------------------------------------ cameras
def look_at_c2w(cam_pos, target=(0, 0, 0), up=(0, 0, 1)):
    """NeRF/blender convention: camera looks along -z, +y is up.  Returns a row-major 4x4."""
    c = np.asarray(cam_pos, np.float64)
    back = c - np.asarray(target, np.float64)
    back /= np.linalg.norm(back)
    right = np.cross(np.asarray(up, np.float64), back)
    right /= np.linalg.norm(right)
    upv = np.cross(back, right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, upv, back, c
    return m


def orbit_poses(n=200, radius=4.0311, elev_deg=(30.0, 20.0)):
    """n camera-to-world matrices on an upper-hemisphere orbit (deterministic, no RNG)."""
    out = []
    for i in range(n):
        az = 2.0 * np.pi * i / n
        el = np.deg2rad(elev_deg[0] + elev_deg[1] * np.sin(4.0 * np.pi * i / n))
        pos = radius * np.array([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)])
        out.append(look_at_c2w(pos))
    return np.stack(out)


def write_transforms_json(path, poses, camera_angle_x=CAMERA_ANGLE_X):
    """Blender schema read by main_headless.cpp:255-272."""
    frames = [{"file_path": "./test/r_%d" % i, "rotation": 0.0,
               "transform_matrix": [[float(v) for v in row] for row in p]} for i, p in enumerate(poses)]
    with open(path, "w") as f:
        json.dump({"camera_angle_x": camera_angle_x, "frames": frames}, f)
    return path


def blender_focal(width, camera_angle_x=CAMERA_ANGLE_X):
    """fx = fy = 0.5f * width / tanf(0.5f * camera_angle_x) (main_headless.cpp:258), in fp32."""
    import ctypes
    tanf = ctypes.CDLL("libm.so.6").tanf  # the C tanf the reference (and volrend_headless) calls
    tanf.restype, tanf.argtypes = ctypes.c_float, [ctypes.c_float]
    a = np.float32(camera_angle_x)
    return float(np.float32(0.5) * np.float32(width) / np.float32(tanf(float(np.float32(0.5) * a))))


def write_tt_dataset(root, poses, fx=1160.0, fy=1160.0, cx=960.0, cy=540.0):
    """TanksAndTemple layout: <root>/intrinsics.txt + <root>/pose/*.txt with 4x4 OpenCV-convention
    c2w matrices (main_headless.cpp:273-297; the loader flips y,z: :373-384)."""
    os.makedirs(os.path.join(root, "pose"), exist_ok=True)
    with open(os.path.join(root, "intrinsics.txt"), "w") as f:
        f.write("%f 0 %f 0\n0 %f %f 0\n0 0 1 0\n0 0 0 1\n" % (fx, cx, fy, cy))
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    for i, p in enumerate(poses):
        np.savetxt(os.path.join(root, "pose", "%06d.txt" % i), p @ flip, fmt="%.8f")
    return os.path.join(root, "pose")


OPT_JSON = {  # renderer/options/opt.json
    "background_brightness": 1.0, "denoise": True, "spp": 6, "enable_probe": False, "grid_max_depth": 4,
    "probe": [0.0, 0.0, 1.0], "probe_disp_size": 100, "show_grid": False, "sigma_thresh": 0.01,
    "step_size": 0.0001, "stop_thresh": 0.01,
}


def write_opt_json(path, **overrides):
    d = dict(OPT_JSON)
    d.update(overrides)
    with open(path, "w") as f:
        json.dump(d, f, indent=2)
    return path
