"""CPU-side checks of the product library: it loads, exports every symbol include/rto.h declares,
and its host logic (options JSON, npz / N3Tree decode, error convention) behaves like the
reference's.  No compute entry point is called here (there is no GPU)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import rt_octree_amd as R
from rt_octree_amd import _lib, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


def _fnv1a64(b):
    h = 1469598103934665603
    for x in np.frombuffer(b, np.uint8).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rto.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rto_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    L = C.CDLL(R.LIB_PATH)
    for name in sorted(declared):
        getattr(L, name)  # AttributeError = missing export
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert R.lib().rto_version().decode().startswith("rt-octree_amd")


def test_library_has_gfx950_code_object():
    blob = open(R.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"render_fast" in blob


def test_options_defaults_and_opt_json(tmp_path):
    o = R.RenderOptions()
    assert (o.step_size, o.sigma_thresh, o.stop_thresh, o.background_brightness) == (
        np.float32(1e-4), np.float32(1e-2), np.float32(1e-2), 1.0)
    assert o.render_bbox == [0, 0, 0, 1, 1, 1] and o.basis_minmax == [0, 24] and o.denoise and o.spp == 1
    p = synth.write_opt_json(str(tmp_path / "opt.json"))
    o = R.RenderOptions.from_json(p)
    assert o.spp == 6 and o.denoise is True and o.probe == [0.0, 0.0, 1.0] and o.grid_max_depth == 4
    assert o.render_bbox == [0, 0, 0, 1, 1, 1]  # not in the JSON binding: keeps its default
    d = dict(synth.OPT_JSON)
    del d["spp"]  # NLOHMANN_DEFINE_TYPE_INTRUSIVE: every key is required
    with pytest.raises(R.RtoError) as e:
        R.RenderOptions.from_json_text(json.dumps(d))
    assert "spp" in str(e.value)
    with pytest.raises(R.RtoError):
        R.RenderOptions.from_json_text("{not json")
    with pytest.raises(R.RtoError):
        R.RenderOptions.from_json(str(tmp_path / "missing.json"))


def _probe(path):
    buf = C.create_string_buffer(2048)
    _lib.check(R.lib().rto_tree_probe_npz(os.fsencode(path), buf, 2048))
    return json.loads(buf.value.decode())


def test_npz_dense_matches_reference_cnpy_listing():
    """our zip/npy reader sees the same bytes the reference's vendored cnpy does (golden listing)."""
    ref = json.load(open(os.path.join(GOLD, "npz_cnpy.json")))["npz_dense.npz"]
    info = _probe(os.path.join(GOLD, "npz_dense.npz"))
    assert info["child_fnv1a64"] == ref["child"]["fnv1a64"]
    assert info["data_fnv1a64"] == ref["data"]["fnv1a64"]
    assert info["capacity"] == ref["child"]["shape"][0] and info["N"] == 2
    assert info["data_format"] == "SH9" and info["data_dim"] == 28 and info["quantized"] == 0
    z = np.load(os.path.join(GOLD, "npz_dense.npz"))
    assert np.allclose(info["scale"], z["invradius3"]) and np.allclose(info["offset"], z["offset"])


def test_npz_quantised_decode():
    """n3tree.cpp:279-340 decode vs an independent numpy statement of SURVEY appendix B 15."""
    path = os.path.join(GOLD, "npz_quant.npz")
    z = np.load(path)
    ref = json.load(open(os.path.join(GOLD, "npz_cnpy.json")))["npz_quant.npz"]
    for k in ("quant_map", "quant_colors", "sigma", "data_retained", "child"):
        assert _fnv1a64(np.ascontiguousarray(z[k]).tobytes()) == ref[k]["fnv1a64"]  # numpy == cnpy bytes
    qm, qc, sg, rt = z["quant_map"], z["quant_colors"], z["sigma"], z["data_retained"]
    nq, cap = qm.shape[0], qm.shape[1]
    nr = rt.shape[0]
    nb = nq + nr
    n_child = cap * 8
    data = np.zeros((n_child, 28), np.float16)
    qmf = qm.reshape(nq, n_child)
    for j in range(nq):
        col = qc[j][qmf[j]]  # [n_child, 3]
        for c in range(3):
            data[:, (j + nr) + c * nb] = col[:, c]
    rtf = rt.reshape(nr, n_child, 3)
    for j in range(nr):
        for c in range(3):
            data[:, j + c * nb] = rtf[j, :, c]
    data[:, 27] = sg.reshape(-1)
    info = _probe(path)
    assert info["quantized"] == 1 and info["capacity"] == cap and info["data_format"] == "SH9"
    assert info["data_fnv1a64"] == _fnv1a64(data.tobytes())
    assert info["child_fnv1a64"] == ref["child"]["fnv1a64"]


def test_npz_stored_and_legacy_format(tmp_path):
    t = synth.make_tree(depth_limit=3, basis_dim=4, seed=3)
    p = str(tmp_path / "t.npz")
    t.save_npz(p, compressed=False)  # STORED members: served straight from the mmap
    info = _probe(p)
    assert info["data_fnv1a64"] == _fnv1a64(t.data.tobytes()) and info["data_format"] == "SH4"
    assert info["max_depth"] == 3
    # legacy file without data_format: SH autodetect (n3tree.cpp:241-254); invradius as f64 scalar
    p2 = str(tmp_path / "legacy.npz")
    np.savez(p2, data_dim=np.int64(13), invradius=np.float64(0.25), offset=t.offset, child=t.child, data=t.data)
    info = _probe(p2)
    assert info["data_format"] == "SH4" and np.allclose(info["scale"], [0.25] * 3)


def test_npz_error_convention(tmp_path):
    t = synth.make_tree(depth_limit=2, basis_dim=4, seed=3)
    buf = C.create_string_buffer(2048)
    L = R.lib()
    assert L.rto_tree_probe_npz(os.fsencode(str(tmp_path / "nope.npz")), buf, 2048) == -5  # RTO_E_IO
    bad = str(tmp_path / "f32.npz")
    np.savez(bad, data_dim=np.int64(13), data_format=np.array("SH4"), invradius3=t.scale, offset=t.offset,
             child=t.child, data=t.data.astype(np.float32))
    assert L.rto_tree_probe_npz(os.fsencode(bad), buf, 2048) == -6
    assert b"data must be stored in half precision" in L.rto_last_error()  # n3tree.cpp:345
    trunc = str(tmp_path / "trunc.npz")
    open(trunc, "wb").write(open(bad, "rb").read()[:200])
    assert L.rto_tree_probe_npz(os.fsencode(trunc), buf, 2048) == -6
    cyc = t.child.copy()
    cyc.reshape(-1)[0] = 10 ** 6  # offset out of range
    badc = str(tmp_path / "badchild.npz")
    np.savez(badc, data_dim=np.int64(13), data_format=np.array("SH4"), invradius3=t.scale, offset=t.offset,
             child=cyc, data=t.data)
    assert L.rto_tree_probe_npz(os.fsencode(badc), buf, 2048) == -6
    assert b"out of range" in L.rto_last_error()


def test_no_device_fails_loudly():
    """No GPU in the authoring container: device entry points must return RTO_E_HIP, never a CPU result."""
    if R.lib().rto_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(R.RtoError) as e:
        R.RenderContext(8, 8)
    assert e.value.code == -4
    t = synth.make_tree(depth_limit=2, basis_dim=4, seed=3)
    with pytest.raises(R.RtoError) as e:
        R.N3Tree.from_arrays(t.child, t.data, t.scale, t.offset, "SH4")
    assert e.value.code == -4


def test_synth_transforms_and_focal(tmp_path):
    poses = synth.orbit_poses(7)
    p = synth.write_transforms_json(str(tmp_path / "transforms_test.json"), poses)
    j = json.load(open(p))
    assert len(j["frames"]) == 7 and abs(j["camera_angle_x"] - 0.6911112070083618) < 1e-15
    assert abs(synth.blender_focal(800) - 1111.11) < 0.01  # main_headless.cpp:258 vs the 1111.11 default
    r = poses[0][:3, :3]
    assert np.allclose(r @ r.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.norm(poses[3][:3, 3]), 4.0311)
