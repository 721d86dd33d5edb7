"""An INDEPENDENT float64 statement of what the regular-tracking estimator must converge to.

Test infrastructure.  Nothing here is derived from rt_core.cuh's loop structure, from oracle/ or from
the HIP kernels; it is the volume-rendering equation for a piecewise-constant medium (a PlenOctree),
written down from the paper's model, plus the two modelling facts the reference documents:

  * a ray visits the octree leaf by leaf; after a leaf it re-starts `step_size` (tree units) beyond
    the leaf's exit point, and that extra length counts as optical depth of
    the leaf just left;
  * leaves with sigma <= sigma_thresh are empty space (appendix B 1).

For a ray crossing segments j = 1..n of length d_j (world units) with density s_j and view-dependent
colour c_j:

    tau_j  = s_j * d_j                        optical depth of segment j
    T_j    = exp(-sum_{i<j} tau_i)            transmittance in front of it
    p_j    = T_j * (1 - exp(-tau_j))          probability that a free-flight sample ends in j
    E[rgb] = sum_j p_j c_j + bg * T_{n+1}     (the estimator composites the background)
    E[a]   = 1 - T_{n+1} = 1 - exp(-tau_total)

Tracking draws free-flight distances d = -log(1 - u) (unit-rate exponential in optical depth) and
returns the colour of the segment where the accumulated optical depth first reaches d; SPP such
samples are averaged.  So one sample is the categorical variable X = c_j w.p. p_j, bg w.p. T_{n+1},
and a mean of n independent samples has variance (E[X^2] - E[X]^2) / n per channel.

The colour of a leaf: sigmoid(sum_k Y_k(view dir) * coeff[channel][k]) with Y_k the real spherical
harmonics (Condon-Shortley phase kept), evaluated here with scipy.
"""
import numpy as np

_CORNER_ORDER = "x major: child index = 4*ix + 2*iy + iz"


def real_sh(dirs, basis_dim):
    """Real spherical harmonics Y_0..Y_{basis_dim-1} of unit vectors `dirs` [n,3] (float64), ordered
    (l, m) = (0,0), (1,-1), (1,0), (1,1), (2,-2), ...; from scipy's complex harmonics."""
    import scipy.special as sp
    d = np.asarray(dirs, np.float64)
    theta, phi = np.arccos(np.clip(d[:, 2], -1, 1)), np.arctan2(d[:, 1], d[:, 0])
    out = []
    lmax = int(round(np.sqrt(basis_dim))) - 1
    for l in range(lmax + 1):
        for m in range(-l, l + 1):
            if hasattr(sp, "sph_harm_y"):
                y = sp.sph_harm_y(l, abs(m), theta, phi)
            else:
                y = sp.sph_harm(abs(m), l, phi, theta)
            out.append(y.real if m == 0 else np.sqrt(2.0) * (y.imag if m < 0 else y.real))
    return np.stack(out, 1)[:, :basis_dim]


def rotate_axis_angle(v, aa):
    """v rotated about the axis aa/|aa| by the angle |aa| (rotation-matrix form, float64)."""
    aa = np.asarray(aa, np.float64)
    ang = np.linalg.norm(aa)
    if ang < 1e-6:
        return np.asarray(v, np.float64)
    k = aa / ang
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    return Rm @ np.asarray(v, np.float64)


class Scene:
    """A PlenOctree as plain float64 numpy: child [cap,8] (relative node offsets, 0 = leaf),
    values [cap,8,data_dim]."""

    def __init__(self, child, data, scale, offset, data_format):
        self.child = np.asarray(child).reshape(-1, 8).astype(np.int64)
        self.val = np.asarray(data).astype(np.float64).reshape(self.child.shape[0], 8, -1)
        self.scale = np.asarray(scale, np.float64)
        self.offset = np.asarray(offset, np.float64)
        self.rgba = data_format.upper().startswith("RGBA")
        self.basis_dim = 0 if self.rgba else (self.val.shape[-1] - 1) // 3

    def locate(self, p):
        """Leaf containing the tree-space point p in [0,1)^3 -> (node, child index, cell lo corner, cell size)."""
        node, lo, size = 0, np.zeros(3), 1.0
        while True:
            size *= 0.5
            bits = (p >= lo + size).astype(np.int64)
            ci = int(4 * bits[0] + 2 * bits[1] + bits[2])
            lo = lo + bits * size
            off = self.child[node, ci]
            if off == 0:
                return node, ci, lo, size
            node += off


def ray_segments(scene, origin_world, dir_world, step_size=1e-4, sigma_thresh=1e-2, bbox=(0, 0, 0, 1, 1, 1)):
    """Leaf segments of one ray, front to back: list of (node, child, world length, sigma).  Empty
    leaves (sigma <= sigma_thresh) are skipped.  The ray re-starts step_size beyond each leaf's exit."""
    o = scene.offset + scene.scale * np.asarray(origin_world, np.float64)
    d = scene.scale * np.asarray(dir_world, np.float64)
    world_per_tree = 1.0 / np.linalg.norm(d)  # world length of one tree-space unit along the ray
    d = d * world_per_tree
    lo, hi = np.asarray(bbox[:3], np.float64) + 1e-6, np.asarray(bbox[3:], np.float64) - 1e-6
    tmin, tmax = 0.0, 1e4
    for a in range(3):
        if abs(d[a]) < 1e-12:
            if not (lo[a] <= o[a] <= hi[a]):
                return []
            continue
        t1, t2 = (lo[a] - o[a]) / d[a], (hi[a] - o[a]) / d[a]
        tmin, tmax = max(tmin, min(t1, t2)), min(tmax, max(t1, t2))
    if tmax < 0 or tmin > tmax:
        return []
    segs = []
    t = tmin
    while t < tmax:
        p = np.clip(o + t * d, 0.0, 1.0 - 1e-6)
        node, ci, cell_lo, size = scene.locate(p)
        t_exit = np.inf
        for a in range(3):  # distance to the cell's exit face along the ray
            if d[a] > 1e-12:
                t_exit = min(t_exit, (cell_lo[a] + size - p[a]) / d[a])
            elif d[a] < -1e-12:
                t_exit = min(t_exit, (cell_lo[a] - p[a]) / d[a])
        dt = t_exit + step_size
        sigma = scene.val[node, ci, -1]
        if sigma > sigma_thresh:
            segs.append((node, ci, dt * world_per_tree, sigma))
        t += dt
    return segs


def leaf_colour(scene, node, ci, view_dir, basis_minmax=(0, 24)):
    v = scene.val[node, ci]
    if scene.rgba:
        return v[:3].copy()
    B = scene.basis_dim
    Y = real_sh(np.asarray(view_dir, np.float64)[None], B)[0]
    mask = (np.arange(B) >= basis_minmax[0]) & (np.arange(B) <= basis_minmax[1])
    Y = Y * mask
    return 1.0 / (1.0 + np.exp(-(v[:3 * B].reshape(3, B) @ Y)))


def expected_sample(scene, origin_world, dir_world, bg=1.0, view_dir=None, **kw):
    """Mean and variance of ONE free-flight sample of the ray: (mean[4], var[4]) for r, g, b (background
    composited) and alpha."""
    view_dir = dir_world if view_dir is None else view_dir
    basis_minmax = kw.pop("basis_minmax", (0, 24))
    segs = ray_segments(scene, origin_world, dir_world, **kw)
    T = 1.0
    m1, m2, a = np.zeros(3), np.zeros(3), 0.0
    for node, ci, length, sigma in segs:
        p = T * (1.0 - np.exp(-sigma * length))
        c = leaf_colour(scene, node, ci, view_dir, basis_minmax)
        m1 += p * c
        m2 += p * c * c
        a += p
        T *= np.exp(-sigma * length)
    m1 += bg * T
    m2 += bg * bg * T
    mean = np.concatenate([m1, [a]])
    var = np.concatenate([m2 - m1 * m1, [a * (1.0 - a)]])
    return mean, np.maximum(var, 0.0)


def pinhole_ray(c2w, W, H, fx, fy, x, y):
    """Pixel (x, y) of a pinhole camera looking along -z with +y up, pixel centres at integer coordinates
    measured from (W/2, H/2) (the convention of the reference's headless renderer)."""
    c2w = np.asarray(c2w, np.float64)
    d_cam = np.array([(x - 0.5 * W) / fx, -(y - 0.5 * H) / fy, -1.0])
    d = c2w[:3, :3] @ d_cam
    return c2w[:3, 3], d / np.linalg.norm(d)


def expected_frame(scene, c2w, W, H, fx, fy, bg=1.0, rot_dirs=None, **kw):
    """-> mean [4,H,W], var [4,H,W] of one sample per pixel"""
    mean = np.zeros((4, H, W))
    var = np.zeros((4, H, W))
    for y in range(H):
        for x in range(W):
            o, d = pinhole_ray(c2w, W, H, fx, fy, x, y)
            vd = d if rot_dirs is None else rotate_axis_angle(d, rot_dirs)
            mean[:, y, x], var[:, y, x] = expected_sample(scene, o, d, bg=bg, view_dir=vd, **kw)
    return mean, var
