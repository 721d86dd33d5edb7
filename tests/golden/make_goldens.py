"""Regenerates the committed fixtures in tests/golden/ (authoring container only: it reads
/root/reference; nothing under tests/ does so at test time).

  pcg32_kat.json        from the reference's own pcg32.h       (oracle/ref_kat/Makefile `golden`)
  guidance_golden.npz   GuidanceNet(8,32,5,2,4) of the imported reference module denoiser/network.py:
                        state_dict + input -> (weight_map, guidance_map), full and compact
  npz_dense.npz / npz_quant.npz + npz_cnpy.json
                        trees written by numpy in the svox key schema, and what the reference's
                        vendored cnpy reads from them (oracle/_ref/cnpy_dump)
  ts_ref_format.ts      a ts module written by the reference's OWN exporter (compact_and_compile + torch.jit.save):
                        a traced closure, conv weights as graph constants -- what volrend_headless must recognise
  frames_golden.npz     tiny frames from the CPU oracle (det math): tree arrays, poses, aux, rgba8
  kat_golden.npz        regression vectors from the CPU oracle: octree point
                        queries incl. faces / corners / the clamp edge, SH basis bit patterns for
                        SH4/9/16/25, the L = 4 filter on a 48x40 image with its saved tensors and gradients

The reference module `denoiser/network.py` tries to JIT-compile its CUDA extension when
`_denoiser` is not importable (network.py:7-47).  An EMPTY placeholder module is registered under
that name so the import succeeds; none of its functions exist or are called (the CUDA filter cannot
run here) -- only the pure-PyTorch network classes are exercised.
"""
import json
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def guidance_golden():
    import torch
    sys.modules.setdefault("_denoiser", types.ModuleType("_denoiser"))
    sys.path.insert(0, REF)
    from denoiser import network as refnet
    torch.manual_seed(0)
    model = refnet.GuidanceNet(8, 32, 5, 2, 4).eval()
    aux = torch.rand(1, 8, 24, 20)
    with torch.no_grad():
        w_full, g_full = model(aux)
        compact = refnet.GuidanceNetCompact(model).eval()
        w_c, g_c = compact(aux)
    out = {"aux": aux.numpy(), "weight_full": w_full.numpy(), "guidance_full": g_full.numpy(),
           "weight_compact": w_c.numpy(), "guidance_compact": g_c.numpy()}
    for k, v in model.state_dict().items():
        out["sd." + k] = v.numpy()
    for k, v in compact.state_dict().items():
        out["csd." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "guidance_golden.npz"), **out)
    print("guidance_golden.npz: %d tensors" % len(out))


def ts_ref_format():
    """A ts_*.ts exactly as the reference's exporter writes it: `compact_and_compile` (denoiser/network.py:170-208)
    traces a closure, so the file is a parameter-less module whose conv weights are graph constants, and
    `runner.compact` saves it with torch.jit.save (runner.py:171-175).  Traced on the CPU (no GPU here);
    loading it with a device argument relocates the constants.  The weights are the trained
    rt-octree_amd/weights/guidance_synth_lego.pt."""
    import torch
    sys.modules.setdefault("_denoiser", types.ModuleType("_denoiser"))
    sys.path.insert(0, REF)
    from denoiser import network as refnet
    model = refnet.GuidanceNet(8, 32, 5, 2, 4).eval()
    model.load_state_dict(torch.load(os.path.join(ROOT, "rt-octree_amd", "weights", "guidance_synth_lego.pt"), map_location="cpu"))
    ts = refnet.compact_and_compile(model, torch.device("cpu"))
    out = os.path.join(HERE, "ts_ref_format.ts")
    torch.jit.save(ts, out)
    print("ts_ref_format.ts: %d bytes" % os.path.getsize(out))


def npz_goldens():
    from rt_octree_amd import synth
    tree = synth.make_tree(depth_limit=4, basis_dim=9, seed=1)
    dense = os.path.join(HERE, "npz_dense.npz")
    tree.save_npz(dense, compressed=True)
    # quantised variant (compress_octree.py schema; decode n3tree.cpp:279-340): n_retain = 7
    rs = np.random.RandomState(2)
    cap = tree.capacity
    n_basis, n_retain = 9, 7
    nq = n_basis - n_retain
    # a compressible codebook (the decode indexes it with a fixed 65536*3 stride per basis)
    quant_colors = ((np.arange(nq * 65536 * 3) % 997) / 997.0 - 0.5).astype(np.float16).reshape(nq, 65536, 3)
    quant_map = rs.randint(0, 65536, (nq, cap, 2, 2, 2)).astype(np.uint16)
    sigma = tree.data[..., -1].copy()
    retained = rs.randn(n_retain, cap, 2, 2, 2, 3).astype(np.float16)
    quant = os.path.join(HERE, "npz_quant.npz")
    np.savez_compressed(quant, data_dim=np.int64(28), data_format=np.array("SH9"), invradius3=tree.scale,
                        offset=tree.offset, child=tree.child, quant_colors=quant_colors, quant_map=quant_map,
                        sigma=sigma, data_retained=retained)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref_kat"), "-s"])
    dump = os.path.join(ROOT, "oracle", "_ref", "cnpy_dump")
    listing = {os.path.basename(p): json.loads(subprocess.check_output([dump, p])) for p in (dense, quant)}
    with open(os.path.join(HERE, "npz_cnpy.json"), "w") as f:
        json.dump(listing, f, indent=1, sort_keys=True)
    print("npz goldens:", {k: sorted(v) for k, v in listing.items()})


def frames_golden():
    import orc
    from rt_octree_amd import synth
    out = {}
    poses = synth.orbit_poses(3)
    out["poses"] = poses
    W, H = 48, 40
    fx = synth.blender_focal(W)
    out["size_fx"] = np.array([W, H, fx], np.float64)
    for name, bd in (("sh9", 9), ("sh16", 16)):
        tree = synth.make_tree(depth_limit=5, basis_dim=bd, seed=21 + bd)
        out[name + ".child"], out[name + ".data"] = tree.child, tree.data
        out[name + ".scale"], out[name + ".offset"] = tree.scale, tree.offset
        ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
        for spp in (1, 6):
            for pi in range(3):
                cam = orc.camera(W, H, fx, fx, poses[pi][:3, :4].T.reshape(-1))
                aux, rgba, st = orc.render_frame(ht, cam, orc.default_options(spp=spp), orc.rng(frame=100 + pi))
                key = "%s.spp%d.pose%d" % (name, spp, pi)
                out[key + ".aux"] = aux
                out[key + ".rgba8"] = orc.rgba8(rgba)
                out[key + ".stats"] = np.array([st[k] for k in ("rays", "rays_in_box", "steps", "levels", "hit_leaves", "hit_rays")], np.int64)
    np.savez_compressed(os.path.join(HERE, "frames_golden.npz"), **out)
    print("frames_golden.npz: %d arrays" % len(out))


def kat_golden():
    import ctypes as C
    import orc
    from rt_octree_amd import synth
    out = {}
    # G3: query_single_from_root on a depth-6 tree
    tree = synth.make_tree(depth_limit=6, basis_dim=4, seed=77)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    rs = np.random.RandomState(5)
    pts = rs.rand(1000, 3).astype(np.float32)
    edge = np.array([0.0, 1.0, 0.5, 0.25, 1.0 - 1e-6, 1.0 - 1e-7, -0.1, 1.3, 0.5 - 2.0 ** -20, 0.5 + 2.0 ** -20], np.float32)
    grid = np.stack(np.meshgrid(edge[:6], edge[2:8], edge[4:], indexing="ij"), -1).reshape(-1, 3)[:300]
    pts = np.concatenate([pts, grid.astype(np.float32)], 0)
    leaf, cube, local, lev = [], [], [], []
    for p3 in pts:
        xyz = (C.c_float * 3)(*[float(v) for v in p3])
        cs, lv = C.c_float(0), C.c_int(0)
        leaf.append(orc.lib().orc_query(C.byref(ht.c), xyz, C.byref(cs), C.byref(lv)))
        cube.append(cs.value)
        local.append(list(xyz))
        lev.append(lv.value)
    out["q.child"], out["q.data"], out["q.scale"], out["q.offset"] = tree.child, tree.data, tree.scale, tree.offset
    out["q.points"] = pts
    out["q.leaf"], out["q.cube_sz"] = np.array(leaf, np.int64), np.array(cube, np.float32)
    out["q.local"], out["q.levels"] = np.array(local, np.float32), np.array(lev, np.int32)
    # G4: maybe_precalc_basis bit patterns
    dirs = rs.randn(256, 3).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    dirs[:6] = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], np.float32)
    out["sh.dirs"] = dirs
    for bd in (4, 9, 16, 25):
        vals = np.zeros((256, bd), np.float32)
        for i, d in enumerate(dirs):
            buf = (C.c_float * 25)()
            orc.lib().orc_sh_basis(bd, (C.c_float * 3)(*[float(v) for v in d]), buf)
            vals[i] = np.array(buf[:bd], np.float32)
        out["sh.basis%d" % bd] = vals
    # G9: the filter, L = 4, 48x40, with the training-side tensors
    L, H, W = 4, 40, 48
    weight = rs.rand(L, H, W).astype(np.float32)
    weight /= weight.sum(0, keepdims=True)
    guidance = (rs.rand(L, H, W) * 6).astype(np.float32)
    noisy = rs.rand(H, W, 4).astype(np.float32)
    noisy[..., 3] = 1
    grad_out = rs.randn(H, W, 4).astype(np.float32)
    img, rf, mx, inv = orc.filter_train_forward(weight, guidance, noisy)
    gw, gg = orc.filter_backward(grad_out, noisy, weight, guidance, rf, mx, inv)
    for k, v in (("weight", weight), ("guidance", guidance), ("noisy", noisy), ("grad_out", grad_out), ("out", img),
                 ("rgb_filtered", rf), ("max_map", mx), ("inv_kernel_sum", inv), ("grad_weight", gw), ("grad_guidance", gg)):
        out["f." + k] = v
    np.savez_compressed(os.path.join(HERE, "kat_golden.npz"), **out)
    print("kat_golden.npz: %d arrays" % len(out))


if __name__ == "__main__":
    which = sys.argv[1:] or ["guidance", "npz", "frames", "kat", "ts"]
    if "ts" in which:
        ts_ref_format()
    if "kat" in which:
        kat_golden()
    if "guidance" in which:
        guidance_golden()
    if "npz" in which:
        npz_goldens()
    if "frames" in which:
        frames_golden()
