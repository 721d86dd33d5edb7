"""The oracle against every pinned vector we have (CPU only):
  * pcg32: tests/golden/pcg32_kat.json, generated from the reference's own pcg32.h;
  * sample_dst<6>: SURVEY.md section 8c G2 (libm logf), and sortedness for every SPP;
  * deterministic logf/expf: <= 0.5000001 ulp of the float64 result, i.e. correctly rounded on the
    sampled inputs; libm frames differ from det frames by a negligible PSNR (tier 2);
  * half decode: all 65536 bit patterns vs numpy;
  * query: every digit/level/local coordinate vs an independent integer-arithmetic walk;
  * SH basis: vs scipy's real spherical harmonics (tolerance 2e-6: fp32 rounding of fp64 products);
  * frames: tests/golden/frames_golden.npz (regression pin of the restatement)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "pcg32_kat.json")))


def test_pcg32_seed_and_stream():
    r = orc.rng(KAT["seed"])
    assert "%016x" % r.state == KAT["state0"] and "%016x" % r.inc == KAT["inc"]
    q = orc.Pcg32(r.state, r.inc)
    assert [orc.lib().orc_pcg32_next_uint(C.byref(q)) for _ in range(16)] == KAT["next_uint"]
    q = orc.Pcg32(r.state, r.inc)
    got = [np.float32(orc.lib().orc_pcg32_next_float(C.byref(q))).view(np.uint32) for _ in range(16)]
    assert [int(x) for x in got] == KAT["next_float_bits"]


def test_pcg32_advance():
    for e in KAT["advance"]:
        q = orc.rng(KAT["seed"])
        orc.lib().orc_pcg32_advance(C.byref(q), e["delta"])
        assert "%016x" % q.state == e["state"], e
        assert orc.lib().orc_pcg32_next_uint(C.byref(q)) == e["next_uint"]


def test_pcg32_frame_and_pixel_jumps():
    for e in KAT["frame_pixel"]:
        q = orc.rng(KAT["seed"], frame=e["frame"])
        orc.lib().orc_pcg32_advance(C.byref(q), e["idx"] * e["spp"])
        assert "%016x" % q.state == e["state"], e
        f = np.float32(orc.lib().orc_pcg32_next_float(C.byref(q)))
        assert int(f.view(np.uint32)) == e["next_float_bits"]


def test_sample_dst_survey_g2_libm():
    L = orc.lib()
    L.orc_set_math_mode(orc.MATH_LIBM)
    try:
        r = orc.rng()
        dst = (C.c_float * 7)()
        L.orc_sample_dst(6, C.byref(r), dst)
    finally:
        L.orc_set_math_mode(orc.MATH_DET)
    want = ["0x1.47e196p-3", "0x1.47125ep-2", "0x1.a4d2a6p-2", "0x1.f0a5d8p-2", "0x1.05ddeap+1", "0x1.775fb6p+1"]
    assert [float(x) for x in dst[:6]] == [float.fromhex(w) for w in want]
    assert dst[6] == np.finfo(np.float32).max


@pytest.mark.parametrize("spp", [1, 2, 3, 4, 6, 8, 16, 32])
def test_sample_dst_sorted_and_consumes_spp_draws(spp):
    L = orc.lib()
    r = orc.rng(frame=5)
    L.orc_pcg32_advance(C.byref(r), 12345 * spp)
    q = orc.Pcg32(r.state, r.inc)
    dst = (C.c_float * (spp + 1))()
    L.orc_sample_dst(spp, C.byref(r), dst)
    v = np.array(dst[:spp], np.float32)
    assert np.all(np.diff(v) >= 0) and np.all(v >= 0)
    draws = sorted(-L.orc_det_logf(np.float32(1.0) - np.float32(L.orc_pcg32_next_float(C.byref(q)))) for _ in range(spp))
    assert [float(np.float32(d)) for d in draws] == [float(x) for x in v]
    assert q.state == r.state  # exactly spp draws


def _ulp_err(f32, ref64):
    u = np.spacing(np.abs(ref64).astype(np.float32)).astype(np.float64)
    return np.abs(f32.astype(np.float64) - ref64) / u


def test_det_logf_expf_accuracy():
    L = orc.lib()
    rs = np.random.RandomState(0)
    xs = np.concatenate([rs.uniform(2 ** -23, 1, 40000), 1 - np.arange(0, 4000) * 2.0 ** -23,
                         2.0 ** rs.uniform(-120, 120, 8000)]).astype(np.float32)
    lg = np.array([L.orc_det_logf(float(x)) for x in xs], np.float32)
    assert _ulp_err(lg, np.log(xs.astype(np.float64))).max() <= 0.5000001
    xe = np.concatenate([rs.uniform(-87, 88, 40000), rs.uniform(-2, 2, 10000)]).astype(np.float32)
    ex = np.array([L.orc_det_expf(float(x)) for x in xe], np.float32)
    assert _ulp_err(ex, np.exp(xe.astype(np.float64))).max() <= 0.5000001
    assert L.orc_det_expf(89.0) == np.inf and L.orc_det_expf(-104.0) == 0.0
    assert L.orc_det_logf(1.0) == 0.0 and L.orc_det_logf(0.0) == -np.inf
    assert L.orc_det_expf(-103.0) == np.float32(np.exp(-103.0))  # subnormal result


def test_half_decode_exhaustive():
    L = orc.lib()
    bits = np.arange(65536, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    got = np.array([L.orc_half2float(int(b)) for b in bits], np.float32)
    ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert ok.all()


def _walk_int(child, p):
    """independent descent on integers: digits are the bits of floor(p * 2^24)"""
    p = np.clip(np.float32(p), np.float32(0), np.float32(1) - np.float32(1e-6)).astype(np.float32)
    ip = (p.astype(np.float64) * 2 ** 24).astype(np.int64)
    node, lvl = 0, 0
    flat = child.reshape(-1, 8)
    while True:
        sh = 23 - lvl
        ci = (((ip[0] >> sh) & 1) << 2) | (((ip[1] >> sh) & 1) << 1) | ((ip[2] >> sh) & 1)
        c = flat[node, ci]
        if c == 0:
            cube = 2.0 ** (lvl + 1)
            loc = p.astype(np.float64) * cube
            return node * 8 + ci, lvl + 1, (loc - np.floor(loc)).astype(np.float32)
        node += c
        lvl += 1


def test_query_matches_integer_walk(small_tree_sh9):
    t = small_tree_sh9
    ht = orc.HostTree(t.child, t.data, t.scale, t.offset, t.data_format)
    rs = np.random.RandomState(1)
    pts = np.concatenate([rs.uniform(-0.05, 1.05, (600, 3)),
                          rs.randint(0, 65, (200, 3)) / 64.0,                       # cell faces / corners
                          np.array([[1 - 1e-6] * 3, [1.0] * 3, [0.0] * 3, [0.5, 0.5, 0.5], [1e-30, 0.25, 0.75]])])
    for p in pts.astype(np.float32):
        xyz = (C.c_float * 3)(*p)
        cube, lv = C.c_float(0), C.c_int(0)
        leaf = orc.lib().orc_query(C.byref(ht.c), xyz, C.byref(cube), C.byref(lv))
        slot, levels, loc = _walk_int(t.child, p)
        assert leaf == slot and lv.value == levels and cube.value == 2.0 ** levels
        assert np.array_equal(np.array(xyz[:], np.float32).view(np.uint32), loc.view(np.uint32)), p


def test_sh_basis_vs_scipy():
    sp = pytest.importorskip("scipy.special")
    rs = np.random.RandomState(2)
    d = rs.randn(128, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    theta, phi = np.arccos(d[:, 2]), np.arctan2(d[:, 1], d[:, 0])
    real = []
    for l in range(5):
        for m in range(-l, l + 1):
            # Condon-Shortley phase kept (the google/spherical-harmonics convention of lumisphere.hpp:39-40)
            if hasattr(sp, "sph_harm_y"):
                y = sp.sph_harm_y(l, abs(m), theta, phi)
            else:
                y = sp.sph_harm(abs(m), l, phi, theta)
            real.append(y.real if m == 0 else np.sqrt(2) * (y.imag if m < 0 else y.real))
    real = np.stack(real, 1)
    for i, v in enumerate(d.astype(np.float32)):
        for bd in (4, 9, 16, 25):
            out = (C.c_float * 25)()
            orc.lib().orc_sh_basis(bd, (C.c_float * 3)(*v), out)
            assert np.allclose(np.array(out[:bd]), real[i, :bd], atol=3e-6), (bd, i)


def test_frames_match_golden():
    """SELF-REGRESSION pin: frames_golden.npz was produced by this repository's oracle (tests/golden/make_goldens.py),
    not by the reference -- it guards the restatement against silent drift.  What ties the restatement to the
    reference's behaviour: the file:line restatement itself, the reference-header pcg32 KAT, and the independent
    rendering-equation model of tests/test_expectation.py."""
    g = np.load(os.path.join(HERE, "golden", "frames_golden.npz"))
    W, H, fx = g["size_fx"]
    W, H = int(W), int(H)
    poses = g["poses"]
    for name, fmt in (("sh9", "SH9"), ("sh16", "SH16")):
        ht = orc.HostTree(g[name + ".child"], g[name + ".data"], g[name + ".scale"], g[name + ".offset"], fmt)
        for spp in (1, 6):
            for pi in range(3):
                cam = orc.camera(W, H, fx, fx, poses[pi][:3, :4].T.reshape(-1))
                aux, rgba, st = orc.render_frame(ht, cam, orc.default_options(spp=spp), orc.rng(frame=100 + pi))
                key = "%s.spp%d.pose%d" % (name, spp, pi)
                assert np.array_equal(aux.view(np.uint32), g[key + ".aux"].view(np.uint32)), key
                assert np.array_equal(orc.rgba8(rgba), g[key + ".rgba8"]), key
                assert [st[k] for k in ("rays", "rays_in_box", "steps", "levels", "hit_leaves", "hit_rays")] == list(g[key + ".stats"])


def test_estimator_properties(small_tree_sh9):
    """alpha in {0,1/SPP,..,1}; aux squares; libm vs det math frames are the same up to rare ulps."""
    t = small_tree_sh9
    ht = orc.HostTree(t.child, t.data, t.scale, t.offset, t.data_format)
    from rt_octree_amd import synth
    W = H = 64
    fx = synth.blender_focal(W)
    cam = orc.camera(W, H, fx, fx, synth.orbit_poses(4)[2][:3, :4].T.reshape(-1))
    aux, rgba, st = orc.render_frame(ht, cam, orc.default_options(spp=6), orc.rng())
    a6 = aux[3] * 6
    assert np.all(np.abs(a6 - np.round(a6)) < 1e-5) and aux[3].max() <= 1 and st["hit_rays"] > 50
    assert np.array_equal(aux[4:], aux[:4] * aux[:4])
    assert np.all(rgba[..., 3] == 1) and np.all(rgba[..., :3][aux[3] == 0] == 1)
    orc.lib().orc_set_math_mode(orc.MATH_LIBM)
    try:
        aux2, _, _ = orc.render_frame(ht, cam, orc.default_options(spp=6), orc.rng())
    finally:
        orc.lib().orc_set_math_mode(orc.MATH_DET)
    mse = np.mean((aux2[:3].astype(np.float64) - aux[:3]) ** 2)
    assert mse < 1e-9  # > 90 dB: the two logf/expf definitions are interchangeable for the image


def test_filter_fp32_exp_accuracy():
    """orc_fexp (the filter's fp32-only, fma-based exp): <= 1 ulp, exact at 0, zero from -87.68 down"""
    L = orc.lib()
    rs = np.random.RandomState(3)
    xs = np.concatenate([rs.uniform(-87.3, 0, 60000), rs.uniform(0, 88.7, 10000), rs.uniform(-1, 1, 10000)]).astype(np.float32)
    got = np.array([L.orc_fexp(float(x)) for x in xs], np.float32)
    assert _ulp_err(got, np.exp(xs.astype(np.float64))).max() <= 1.0
    assert L.orc_fexp(0.0) == 1.0 and L.orc_fexp(-87.7) == 0.0 and L.orc_fexp(-3.0e38) == 0.0
    assert 0.0 < L.orc_fexp(-87.5) < 1.2e-38  # subnormal just below FLT_MIN, then exactly 0
    assert L.orc_fexp(88.8) == np.inf


def test_unsupported_inputs():
    from rt_octree_amd import synth
    t = synth.make_tree(depth_limit=3, basis_dim=4, seed=1)
    ht = orc.HostTree(t.child, t.data, t.scale, t.offset, t.data_format)
    cam = orc.camera(8, 8, 10, 10, synth.orbit_poses(1)[0][:3, :4].T.reshape(-1))
    with pytest.raises(RuntimeError):
        orc.render_frame(ht, cam, orc.default_options(spp=5), orc.rng())


def test_oracle_reproduces_committed_kat_vectors():
    """tests/golden/kat_golden.npz: point queries, SH basis bit patterns, the
    L = 4 filter with its saved tensors and gradients -- SELF-REGRESSION vectors (generated by this oracle): they pin the
    oracle against silent drift, not against the reference."""
    import ctypes as C
    g = np.load(os.path.join(HERE, "golden", "kat_golden.npz"))
    ht = orc.HostTree(g["q.child"], g["q.data"], g["q.scale"], g["q.offset"], "SH4")
    for i, p3 in enumerate(g["q.points"]):
        xyz = (C.c_float * 3)(*[float(v) for v in p3])
        cs, lv = C.c_float(0), C.c_int(0)
        leaf = orc.lib().orc_query(C.byref(ht.c), xyz, C.byref(cs), C.byref(lv))
        assert leaf == g["q.leaf"][i] and lv.value == g["q.levels"][i]
        assert np.float32(cs.value).view(np.uint32) == g["q.cube_sz"][i].view(np.uint32)
        assert np.array_equal(np.array(list(xyz), np.float32).view(np.uint32), g["q.local"][i].view(np.uint32)), i
    for bd in (4, 9, 16, 25):
        want = g["sh.basis%d" % bd]
        for i, d in enumerate(g["sh.dirs"]):
            buf = (C.c_float * 25)()
            orc.lib().orc_sh_basis(bd, (C.c_float * 3)(*[float(v) for v in d]), buf)
            assert np.array_equal(np.array(buf[:bd], np.float32).view(np.uint32), want[i].view(np.uint32)), (bd, i)
    out, rf, mx, inv = orc.filter_train_forward(g["f.weight"], g["f.guidance"], g["f.noisy"])
    gw, gg = orc.filter_backward(g["f.grad_out"], g["f.noisy"], g["f.weight"], g["f.guidance"], rf, mx, inv)
    for name, got in (("out", out), ("rgb_filtered", rf), ("max_map", mx), ("inv_kernel_sum", inv),
                      ("grad_weight", gw), ("grad_guidance", gg)):
        assert np.array_equal(got.view(np.uint32), g["f." + name].view(np.uint32)), name
