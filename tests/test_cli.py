"""volrend_headless (C++ host over the C ABI): the reference CLI contract of
renderer/main_headless.cpp -- tree.npz + transforms_test.json + opt.json (+ ts_*.ts) in,
r_<i>.png / buf_r_<i>.bin + the 5-line timing report out."""
import os
import re
import subprocess

import numpy as np
import pytest

import rt_octree_amd as R
from rt_octree_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(os.path.dirname(HERE), "rt-octree_amd", "bin", "volrend_headless")


def _run(args, **kw):
    return subprocess.run([BIN] + args, capture_output=True, text=True, timeout=600, **kw)


def test_cli_help_and_usage_errors(tmp_path):
    assert os.path.exists(BIN), "volrend_headless was not built (make -C rt-octree_amd/csrc)"
    r = _run(["--help"])
    assert r.returncode == 0 and "--ts_module" in r.stdout and "--dataset" in r.stdout
    assert _run([]).returncode == 1  # needs npz_file and poses
    p = synth.write_transforms_json(str(tmp_path / "t.json"), synth.orbit_poses(2))
    r = _run([str(tmp_path / "missing.npz"), p])
    assert r.returncode == 1 and "does not exist" in (r.stdout + r.stderr)
    r = _run([str(tmp_path / "missing.npz"), str(tmp_path / "nope.json")])
    assert r.returncode == 1 and "cannot open poses file" in r.stderr
    r = _run([str(tmp_path / "x.npz"), p, "--dataset", "bogus"])
    assert r.returncode == 1 and "unknown dataset type" in r.stderr


def _scene(tmp_path, n=3):
    tree = synth.make_tree(depth_limit=6, basis_dim=9, seed=7)
    tp = tree.save_npz(str(tmp_path / "tree.npz"))
    poses = synth.orbit_poses(n)
    pp = synth.write_transforms_json(str(tmp_path / "transforms_test.json"), poses)
    return tree, tp, poses, pp


@pytest.mark.gpu
def test_cli_blender_png_matches_oracle(tmp_path):
    """no denoise: the PNG bytes decode to exactly the oracle's RGBA8 for every pose (RNG advanced
    warmup + i times, main_headless.cpp:469-506)."""
    import orc
    from PIL import Image
    tree, tp, poses, pp = _scene(tmp_path)
    op = synth.write_opt_json(str(tmp_path / "opt.json"), denoise=False, spp=6)
    out = str(tmp_path / "out")
    r = _run([tp, pp, "--options", op, "--dataset", "blender", "-w", "96", "-h", "64", "-o", out, "--warmup", "5"])
    assert r.returncode == 0, r.stderr
    assert re.search(r"render: [0-9.]+ ms per frame\ntorch:  [0-9.]+ ms per frame\nfilter: [0-9.]+ ms per frame\nall:    [0-9.]+ ms per frame\nFPS:    [0-9.]+", r.stdout)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    fx = synth.blender_focal(96)
    for i in range(3):
        got = np.array(Image.open(os.path.join(out, "r_%d.png" % i)))
        cam = orc.camera(96, 64, fx, fx, poses[i][:3, :4].T.reshape(-1))
        _, rgba, _ = orc.render_frame(ht, cam, orc.default_options(spp=6), orc.rng(frame=5 + i))
        assert got.shape == (64, 96, 4) and np.array_equal(got, orc.rgba8(rgba)), i


@pytest.mark.gpu
def test_cli_write_buffer_and_shard(tmp_path):
    """--write_buffer dumps aux [8,H,W] fp32 (dataset.py:161-163 layout); --shard i/N renders a subset
    with unchanged per-frame RNG, so shards reproduce the unsharded frames."""
    import orc
    tree, tp, poses, pp = _scene(tmp_path, n=4)
    op = synth.write_opt_json(str(tmp_path / "opt.json"), denoise=False, spp=2)
    full, s1 = str(tmp_path / "full"), str(tmp_path / "s1")
    base = [tp, pp, "--options", op, "-w", "48", "-h", "40", "--warmup", "2", "--write_buffer"]
    assert _run(base + ["-o", full]).returncode == 0
    assert _run(base + ["-o", s1, "--shard", "1/2"]).returncode == 0
    assert sorted(os.listdir(s1)) == ["buf_r_1.bin", "buf_r_3.bin"]
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    fx = synth.blender_focal(48)
    for i in range(4):
        buf = np.fromfile(os.path.join(full, "buf_r_%d.bin" % i), np.float32).reshape(8, 40, 48)
        cam = orc.camera(48, 40, fx, fx, poses[i][:3, :4].T.reshape(-1))
        aux, _, _ = orc.render_frame(ht, cam, orc.default_options(spp=2), orc.rng(frame=2 + i))
        assert np.array_equal(buf.view(np.uint32), aux.view(np.uint32)), i
        if i % 2 == 1:
            assert open(os.path.join(s1, "buf_r_%d.bin" % i), "rb").read() == buf.tobytes()


@pytest.mark.gpu
def test_cli_denoise_with_torchscript_module(tmp_path):
    """full C2-style pipeline through libtorch: ts module exported like compact_and_compile, filter on
    the HIP kernel; the PNG equals the Python host's result for the same inputs."""
    import torch
    from PIL import Image
    from rt_octree_amd import denoiser
    tree, tp, poses, pp = _scene(tmp_path, n=2)
    torch.manual_seed(0)
    full = denoiser.GuidanceNet(8, 32, 5, 2, 4)
    ts = denoiser.compact_and_compile(full, device="cuda:0", example_hw=(64, 80))
    tsp = str(tmp_path / "ts_latest.ts")
    ts.save(tsp)
    op = synth.write_opt_json(str(tmp_path / "opt.json"))  # spp 6, denoise true: the reference's opt.json
    out = str(tmp_path / "out")
    r = _run([tp, pp, "--options", op, "--ts_module", tsp, "-w", "80", "-h", "64", "-o", out, "--warmup", "1", "--torch_net"])
    assert r.returncode == 0, r.stderr
    assert "GuidanceNet runs through libtorch" in r.stdout
    # default: the compact two-layer module is recognised and runs as the fused HIP kernel
    out_f = str(tmp_path / "out_fused")
    rf = _run([tp, pp, "--options", op, "--ts_module", tsp, "-w", "80", "-h", "64", "-o", out_f, "--warmup", "1"])
    assert rf.returncode == 0, rf.stderr
    assert "GuidanceNet runs as the fused HIP kernel" in rf.stdout
    # the same through the Python host
    dt = R.N3Tree(tp)
    ctx = R.RenderContext(80, 64)
    fx = synth.blender_focal(80)
    cam = R.Camera(80, 64, fx, fx)
    dn = denoiser.Denoiser(tsp)
    opt = R.RenderOptions.from_json(op)
    for i in range(2):
        cam.set_c2w(poses[i])
        ctx.rng_seed()
        ctx.rng_advance((1 + i) << 32)
        R.launch_renderer(dt, cam, opt, ctx)
        dn.denoise(cam, ctx)
        want = ctx.download_rgba8()
        got = np.array(Image.open(os.path.join(out, "r_%d.png" % i)))
        assert np.array_equal(got, want), i
        # fused kernel on the same TorchScript weights: the Python FusedGuidanceNet gives the same bytes
        fused = denoiser.FusedGuidanceNet(torch.jit.load(tsp, map_location="cuda:0"))
        wm, gm = fused(torch.as_tensor(ctx.aux_view(), device="cuda:0"))
        R.filtering(None, wm[0].contiguous(), gm[0].contiguous(), ctx.noisy_ptr, ctx.image_ptr)
        got_f = np.array(Image.open(os.path.join(out_f, "r_%d.png" % i)))
        assert np.array_equal(got_f, ctx.download_rgba8()), i
        assert np.abs(got_f.astype(int) - got.astype(int)).max() <= 2  # fp16 accumulation order only
    # denoise = true without --ts_module: the reference's error text
    r = _run([tp, pp, "--options", op, "-w", "80", "-h", "64"])
    assert r.returncode == 1 and "No torchscript module is given to denoiser." in r.stderr


@pytest.mark.gpu
def test_cli_reference_format_ts_takes_the_fused_kernel(tmp_path):
    """A ts_*.ts as the reference's exporter writes it (tests/golden/ts_ref_format.ts: traced closure, weights
    as graph constants, no parameters) runs as the fused HIP kernel, and its PNGs agree with the libtorch
    route of the same file to fp16 accumulation order."""
    from PIL import Image
    tree, tp, poses, pp = _scene(tmp_path, n=2)
    tsp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ts_ref_format.ts")
    op = synth.write_opt_json(str(tmp_path / "opt.json"))
    out_f, out_t = str(tmp_path / "fused"), str(tmp_path / "torch")
    rf = _run([tp, pp, "--options", op, "--ts_module", tsp, "-w", "80", "-h", "64", "-o", out_f, "--warmup", "1"])
    assert rf.returncode == 0, rf.stderr
    assert "GuidanceNet runs as the fused HIP kernel" in rf.stdout
    rt = _run([tp, pp, "--options", op, "--ts_module", tsp, "-w", "80", "-h", "64", "-o", out_t, "--warmup", "1", "--torch_net"])
    assert rt.returncode == 0, rt.stderr
    assert "GuidanceNet runs through libtorch" in rt.stdout
    for i in range(2):
        a = np.array(Image.open(os.path.join(out_f, "r_%d.png" % i))).astype(int)
        b = np.array(Image.open(os.path.join(out_t, "r_%d.png" % i))).astype(int)
        assert np.abs(a - b).max() <= 2, i
    # the Python host recognises the same file
    import torch
    from rt_octree_amd import denoiser
    fused = denoiser.FusedGuidanceNet(torch.jit.load(tsp, map_location="cuda:0"))
    assert fused.c1 == 32 and fused.levels == 4


@pytest.mark.gpu
def test_cli_fast_filter_route(tmp_path):
    """--fast_filter: fused GuidanceNet -> packed fp16 maps -> factorised filter.  PNG bytes within 1 LSB of the
    default (bit-exact filter) route -- the (uint8)(f * 255) truncation turns a 1e-6 difference into a step at most --
    and identical between --batch 1 and the batched loop."""
    from PIL import Image
    tree, tp, poses, pp = _scene(tmp_path, n=3)
    tsp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ts_ref_format.ts")
    op = synth.write_opt_json(str(tmp_path / "opt.json"))
    outs = {}
    for name, extra in (("exact", []), ("fast", ["--fast_filter"]), ("fast1", ["--fast_filter", "--batch", "1"])):
        out = str(tmp_path / name)
        r = _run([tp, pp, "--options", op, "--ts_module", tsp, "-w", "96", "-h", "72", "-o", out, "--warmup", "1"] + extra)
        assert r.returncode == 0, r.stderr
        outs[name] = [np.array(Image.open(os.path.join(out, "r_%d.png" % i))).astype(int) for i in range(3)]
    changed = 0
    for i in range(3):
        assert np.abs(outs["fast"][i] - outs["exact"][i]).max() <= 1, i
        assert np.array_equal(outs["fast"][i], outs["fast1"][i]), i
        changed += int((outs["fast"][i] != outs["exact"][i]).sum())
    print("fast vs exact route: %d of %d PNG bytes differ by one code value" % (changed, 3 * 96 * 72 * 4))
    assert changed < 0.002 * 3 * 96 * 72 * 4, changed  # a handful of LSB flips at most


@pytest.mark.gpu
def test_cli_denoise_cull_same_pngs(tmp_path):
    """the batched --fast_filter loop fills the GuidanceNet / filter tiles that see only culled render tiles: same PNG
    bytes as with --no_denoise_cull, on frames large enough to hold such tiles"""
    from PIL import Image
    tree, tp, poses, pp = _scene(tmp_path, n=3)
    tsp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ts_ref_format.ts")
    op = synth.write_opt_json(str(tmp_path / "opt.json"))
    outs = {}
    for name, extra in (("cull", []), ("all", ["--no_denoise_cull"])):
        out = str(tmp_path / name)
        r = _run([tp, pp, "--options", op, "--ts_module", tsp, "-w", "400", "-h", "304", "-o", out, "--warmup", "1", "--fast_filter"] + extra)
        assert r.returncode == 0, r.stderr
        outs[name] = [np.array(Image.open(os.path.join(out, "r_%d.png" % i))) for i in range(3)]
    for i in range(3):
        assert np.array_equal(outs["cull"][i], outs["all"][i]), i
        assert (outs["cull"][i][..., :3] != 255).any()


def _parse_poses(out):
    lines = [l for l in out.splitlines() if l and not l.startswith("INFO")]
    head = lines[0].split()
    assert head[0] == "POSES"
    n, w, h, fx, fy = int(head[1]), int(head[2]), int(head[3]), float(head[4]), float(head[5])
    names, mats = [], []
    for l in lines[1:1 + n]:
        parts = l.split()
        names.append(parts[0])
        mats.append(np.array([float(x) for x in parts[1:13]], np.float32).reshape(4, 3))  # rows = glm columns
    return n, w, h, fx, fy, names, np.stack(mats)


def test_cli_pose_loader_blender(tmp_path):
    poses = synth.orbit_poses(5)
    pp = synth.write_transforms_json(str(tmp_path / "transforms_test.json"), poses)
    r = _run(["unused.npz", pp, "--print_poses", "-w", "400", "-h", "400"])
    assert r.returncode == 0, r.stderr
    n, w, h, fx, fy, names, m = _parse_poses(r.stdout)
    assert (n, w, h) == (5, 400, 400) and names == ["r_%d" % i for i in range(5)]  # main_headless.cpp:271
    assert abs(fx - synth.blender_focal(400)) < 1e-4 and fx == fy
    for i in range(5):
        assert np.allclose(m[i], poses[i][:3, :4].T.astype(np.float32), atol=1e-6)  # column-major 4x3
    assert "Use NeRF camera convention" in r.stdout
    # `--file` names the tree like the first positional does (opts.cpp:12,36); `--draw` is accepted and ignored
    r2 = _run(["--file", "unused.npz", pp, "--print_poses", "-w", "400", "-h", "400", "--draw", "x.draw.npz"])
    assert r2.returncode == 0 and r2.stdout == r.stdout


def test_cli_gpus_flag_shards_the_poses_over_child_processes(tmp_path):
    """--gpus N: a parent that touches no GPU starts N copies of itself, `--shard i/N` with
    HIP_VISIBLE_DEVICES = GPU i each, relays their output tagged by rank and fails if one of them fails.  Checked here
    without a GPU through --print_poses: every pose printed by exactly one rank, pose i by rank i mod N."""
    poses = synth.orbit_poses(7)
    pp = synth.write_transforms_json(str(tmp_path / "transforms_test.json"), poses)
    r = _run(["unused.npz", pp, "--print_poses", "-w", "400", "-h", "400", "--gpus", "3"])
    assert r.returncode == 0, r.stderr
    seen = {}
    for line in r.stdout.splitlines():
        m = re.match(r"\[rank (\d+)\] (r_(\d+)) (.*)", line)
        if m:
            assert int(m.group(3)) not in seen
            seen[int(m.group(3))] = (int(m.group(1)), np.array([float(x) for x in m.group(4).split()], np.float32))
    assert sorted(seen) == list(range(7))
    for i, (rank, mat) in seen.items():
        assert rank == i % 3
        assert np.allclose(mat.reshape(4, 3), poses[i][:3, :4].T.astype(np.float32), atol=1e-6)
    assert r.stdout.count("POSES 7 400 400") == 3  # every rank parsed the whole file
    # a failing child fails the run; --gpus and --shard exclude each other
    r = _run(["unused.npz", str(tmp_path / "nope.json"), "--print_poses", "--gpus", "2"])
    assert r.returncode == 1
    r = _run(["unused.npz", pp, "--print_poses", "--gpus", "2", "--shard", "0/2"])
    assert r.returncode == 1 and "one of --gpus and --shard" in r.stderr
    # the caller's own device list is dealt out in order
    r = subprocess.run([BIN, "unused.npz", pp, "--print_poses", "--gpus", "2"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="5,3"))
    assert r.returncode == 0 and "[rank 1] r_1 " in r.stdout


@pytest.mark.gpu
def test_cli_gpus_two_ranks_render_the_union(tmp_path):
    """two ranks (sharing the one GPU of the test box: HIP_VISIBLE_DEVICES "0,0") write the PNGs of the unsharded run,
    byte for byte, and the parent prints ONE report for their union"""
    tree, tp, poses, pp = _scene(tmp_path, n=5)
    op = synth.write_opt_json(str(tmp_path / "opt.json"), denoise=False, spp=6)
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    base = [tp, pp, "--options", op, "-w", "64", "-h", "48", "--warmup", "3"]
    assert _run(base + ["-o", one]).returncode == 0
    r = subprocess.run([BIN] + base + ["-o", two, "--gpus", "2"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="0,0"))
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(two)) == sorted(os.listdir(one)) == ["r_%d.png" % i for i in range(5)]
    for f in os.listdir(one):
        assert open(os.path.join(one, f), "rb").read() == open(os.path.join(two, f), "rb").read(), f
    assert re.search(r"\nrender: [0-9.]+ ms per frame\ntorch:  [0-9.]+ ms per frame\nfilter: [0-9.]+ ms per frame\nall:    [0-9.]+ ms per frame\nFPS:    [0-9.]+\nINFO: 5 frames on 2 GPUs", r.stdout)
    assert "RANK_REPORT" not in r.stdout


def test_cli_pose_loader_tt(tmp_path):
    """TanksAndTemple: 1920x1080, ../intrinsics.txt, OpenCV -> NeRF flip (main_headless.cpp:273-297,372-384)"""
    poses = synth.orbit_poses(3)
    pose_dir = synth.write_tt_dataset(str(tmp_path / "tt"), poses, fx=1166.5, fy=1163.25)
    r = _run(["unused.npz", pose_dir, "--dataset", "tt", "--print_poses"])
    assert r.returncode == 0, r.stderr
    n, w, h, fx, fy, names, m = _parse_poses(r.stdout)
    assert (n, w, h) == (3, 1920, 1080) and abs(fx - 1166.5) < 1e-3 and abs(fy - 1163.25) < 1e-3
    assert names == ["000000", "000001", "000002"]
    for i in range(3):  # files hold c2w * diag(1,-1,-1,1); the loader flips back
        assert np.allclose(m[i], poses[i][:3, :4].T.astype(np.float32), atol=1e-5)
    assert "Use OpenCV camera convention" in r.stdout


def test_cli_pose_loader_llff(tmp_path):
    """LLFF poses_bounds.npy: factor 4, axis swap, bounds scaling, recentring (main_headless.cpp:298-390)
    against an independent numpy statement."""
    rs = np.random.RandomState(4)
    n = 6
    pb = np.zeros((n, 17))
    c2w = synth.orbit_poses(n, radius=3.0, elev_deg=(10.0, 5.0))
    for i in range(n):
        m = np.zeros((3, 5))
        m[:, :4] = c2w[i][:3, :4] + rs.randn(3, 4) * 0.01
        m[:, 4] = (3024.0, 4032.0, 3260.0)  # H, W, focal
        pb[i, :15] = m.reshape(-1)
        pb[i, 15:] = (1.2 + 0.1 * i, 9.0 + i)
    d = tmp_path / "llff"
    (d / "images_4").mkdir(parents=True)
    for i in range(n):
        (d / "images_4" / ("img_%03d.png" % i)).write_bytes(b"")
    np.save(str(d / "poses_bounds.npy"), pb)
    r = _run(["unused.npz", str(d / "poses_bounds.npy"), "--dataset", "llff", "--print_poses"])
    assert r.returncode == 0, r.stderr
    cnt, w, h, fx, fy, names, m = _parse_poses(r.stdout)
    assert (cnt, w, h) == (n, 1008, 756) and abs(fx - 815.0) < 1e-3 and names[0] == "img_000"
    # independent statement
    bds_min = pb[:, 15].min()
    mats = []
    for i in range(n):
        t = pb[i, :15].reshape(3, 5)[:, :4].astype(np.float32)           # 3x4 row-major
        t = np.stack([t[:, 1], -t[:, 0], t[:, 2], t[:, 3] * np.float32(1.0 / (bds_min * 0.75))], 1)
        mats.append(t)
    mats = np.stack(mats).astype(np.float64)
    z = mats[:, :, 2].sum(0) / n
    z /= np.linalg.norm(z)
    up = mats[:, :, 1].sum(0) / n
    cen = mats[:, :, 3].sum(0) / n
    x = np.cross(up, z); x /= np.linalg.norm(x)
    y = np.cross(z, x); y /= np.linalg.norm(y)
    avg = np.eye(4); avg[:3, 0], avg[:3, 1], avg[:3, 2], avg[:3, 3] = x, y, z, cen
    inv = np.linalg.inv(avg)
    for i in range(n):
        p4 = np.eye(4); p4[:3, :4] = mats[i]
        want = (inv @ p4)[:3, :4]
        assert np.allclose(m[i], want.T, atol=2e-5), i
    assert "Use LLFF camera convention" in r.stdout


@pytest.mark.gpu
def test_cli_batch_equals_frame_loop(tmp_path):
    """--batch 4 (persistent ray-queue kernel + batched libtorch forward + batched filter) writes the
    same PNG bytes as the reference-style one-frame-per-launch loop."""
    import torch
    from rt_octree_amd import denoiser
    tree, tp, poses, pp = _scene(tmp_path, n=6)
    torch.manual_seed(0)
    ts = denoiser.compact_and_compile(denoiser.GuidanceNet(8, 32, 5, 2, 4), device="cuda:0", example_hw=(48, 64))
    tsp = str(tmp_path / "ts_latest.ts")
    ts.save(tsp)
    op = synth.write_opt_json(str(tmp_path / "opt.json"))
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    base = [tp, pp, "--options", op, "--ts_module", tsp, "-w", "64", "-h", "48", "--warmup", "2"]
    r1 = _run(base + ["-o", a])
    r4 = _run(base + ["-o", b, "--batch", "4"])
    assert r1.returncode == 0 and r4.returncode == 0, r1.stderr + r4.stderr
    assert "FPS:" in r4.stdout
    for i in range(6):
        assert open(os.path.join(a, "r_%d.png" % i), "rb").read() == open(os.path.join(b, "r_%d.png" % i), "rb").read(), i


@pytest.mark.gpu
def test_cli_quant_direct_equals_expanded(tmp_path):
    """--quant_direct (codebook shading, no dense expansion) writes the same PNG bytes as the
    reference's decode-then-render route (n3tree.cpp:279-340), single frames and batches."""
    tree, _, poses, pp = _scene(tmp_path, n=4)
    qp = str(tmp_path / "tree_q.npz")
    tree.save_quant_npz(qp, n_retain=1)
    op = synth.write_opt_json(str(tmp_path / "opt.json"), denoise=False)
    a, b, c = (str(tmp_path / x) for x in "abc")
    base = [qp, pp, "--options", op, "-w", "64", "-h", "48", "--warmup", "2"]
    ra = _run(base + ["-o", a])
    rb = _run(base + ["-o", b, "--quant_direct"])
    rc = _run(base + ["-o", c, "--quant_direct", "--batch", "4"])
    assert ra.returncode == 0 and rb.returncode == 0 and rc.returncode == 0, ra.stderr + rb.stderr + rc.stderr
    for i in range(4):
        ref = open(os.path.join(a, "r_%d.png" % i), "rb").read()
        assert ref == open(os.path.join(b, "r_%d.png" % i), "rb").read(), i
        assert ref == open(os.path.join(c, "r_%d.png" % i), "rb").read(), i


def _oracle_pngs_match(tree, print_out, out_dir, scale, warmup, spp, ndc=None):
    """Renders the poses volrend_headless itself parsed (--print_poses) with the oracle and compares
    the RGBA8 bytes of every PNG; intrinsics scaled like main_headless.cpp:407-417."""
    import orc
    from PIL import Image
    n, w, h, fx, fy, names, mats = _parse_poses(print_out)
    sw, sh = int(w * np.float32(scale)), int(h * np.float32(scale))
    sfx = np.float32(fx) * (np.float32(sw) / np.float32(w))
    sfy = np.float32(fy) * (np.float32(sh) / np.float32(h))
    kw = {} if ndc is None else {"ndc": ndc(w, h, fx)}
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format, **kw)
    for i in range(n):
        got = np.array(Image.open(os.path.join(out_dir, "%s.png" % names[i])))
        cam = orc.camera(sw, sh, float(sfx), float(sfy), mats[i].reshape(-1))
        _, rgba, _ = orc.render_frame(ht, cam, orc.default_options(spp=spp), orc.rng(frame=warmup + i))
        assert got.shape == (sh, sw, 4), (got.shape, sh, sw)
        assert np.array_equal(got, orc.rgba8(rgba)), names[i]
    return n


@pytest.mark.gpu
def test_cli_tt_dataset_end_to_end(tmp_path):
    """TanksAndTemple pose directory + ../intrinsics.txt (1920x1080, OpenCV -> NeRF
    flip) rendered through volrend_headless == the oracle on the same poses, byte for byte."""
    tree = synth.make_tree(depth_limit=6, basis_dim=9, seed=7)
    tp = tree.save_npz(str(tmp_path / "tree.npz"))
    # T&T cameras look at the scene from ~4 units with fx ~ 1160 at 1920x1080
    pose_dir = synth.write_tt_dataset(str(tmp_path / "tt"), synth.orbit_poses(3), fx=1166.5, fy=1163.25)
    op = synth.write_opt_json(str(tmp_path / "opt.json"), denoise=False, spp=4)
    out = str(tmp_path / "out")
    base = [tp, pose_dir, "--dataset", "tt"]
    rp = _run(base + ["--print_poses"])
    r = _run(base + ["--options", op, "--scale", "0.0625", "-o", out, "--warmup", "3"])
    assert rp.returncode == 0 and r.returncode == 0, rp.stderr + r.stderr
    assert _oracle_pngs_match(tree, rp.stdout, out, 0.0625, 3, 4) == 3


@pytest.mark.gpu
def test_cli_llff_dataset_end_to_end(tmp_path):
    """LLFF poses_bounds.npy (factor 4, recentring) + NDC ray warp
    (maybe_world2ndc volrend.cu:35-56 with ndc_width/height/focal set as main_headless.cpp:400-405
    does) through volrend_headless == the oracle, byte for byte."""
    rs = np.random.RandomState(4)
    n = 3
    pb = np.zeros((n, 17))
    c2w = synth.orbit_poses(8, radius=0.6, elev_deg=(5.0, 3.0))
    for i in range(n):
        m = np.zeros((3, 5))
        # LLFF stores [down, right, back] columns; the loader swaps them back (main_headless.cpp:330-340)
        r_, u_, b_, t_ = c2w[i][:3, 0], c2w[i][:3, 1], c2w[i][:3, 2], c2w[i][:3, 3]
        m[:, 0], m[:, 1], m[:, 2], m[:, 3] = -u_, r_, b_, t_ + rs.randn(3) * 0.01
        m[:, 4] = (3024.0, 4032.0, 3260.0)  # H, W, focal
        pb[i, :15] = m.reshape(-1)
        pb[i, 15:] = (1.2 + 0.1 * i, 9.0 + i)
    d = tmp_path / "llff"
    (d / "images_4").mkdir(parents=True)
    for i in range(n):
        (d / "images_4" / ("img_%03d.png" % i)).write_bytes(b"")
    np.save(str(d / "poses_bounds.npy"), pb)
    tree = synth.make_tree(depth_limit=6, basis_dim=9, seed=7)
    tp = tree.save_npz(str(tmp_path / "tree.npz"))
    op = synth.write_opt_json(str(tmp_path / "opt.json"), denoise=False, spp=6)
    out = str(tmp_path / "out")
    base = [tp, str(d / "poses_bounds.npy"), "--dataset", "llff"]
    rp = _run(base + ["--print_poses"])
    r = _run(base + ["--options", op, "--scale", "0.125", "-o", out, "--warmup", "2"])
    assert rp.returncode == 0 and r.returncode == 0, rp.stderr + r.stderr
    got = _oracle_pngs_match(tree, rp.stdout, out, 0.125, 2, 6, ndc=lambda w, h, fx: (float(w), float(h), float(fx)))
    assert got == n
    from PIL import Image
    img = np.array(Image.open(os.path.join(out, "img_000.png")))
    assert img[..., :3].std() > 1.0  # the NDC frustum actually sees the scene (not a flat background)


@pytest.mark.gpu
def test_config_c1_one_pose_400x400_spp1_no_denoiser(tmp_path):
    """BASELINE.json configs[0] at its own size: tree.npz, ONE test pose, 400x400, SPP 1, no denoiser (no --ts_module: the
    reference would abort without one, main_headless.cpp:455-456 -- documented deviation), 100 warm-up frames as the
    reference runs them (main_headless.cpp:469-479).  The CPU N3Tree traversal of that config is the oracle: the CLI's PNG
    must decode to its RGBA8 bytes, through the batched kernels (default) and one launch per frame (--batch 1)."""
    import orc
    from PIL import Image
    tree = synth.make_tree(depth_limit=7, basis_dim=16, seed=20230418)
    tp = tree.save_npz(str(tmp_path / "tree.npz"))
    poses = synth.orbit_poses(200)[:1]
    pp = synth.write_transforms_json(str(tmp_path / "transforms_test.json"), poses)
    op = synth.write_opt_json(str(tmp_path / "opt.json"), denoise=False, spp=1)
    ht = orc.HostTree(tree.child, tree.data, tree.scale, tree.offset, tree.data_format)
    fx = synth.blender_focal(400)
    cam = orc.camera(400, 400, fx, fx, poses[0][:3, :4].T.reshape(-1))
    aux, rgba, st = orc.render_frame(ht, cam, orc.default_options(spp=1), orc.rng(frame=100))
    assert st["rays"] == 160000 and st["hit_rays"] > 5000
    for extra in ([], ["--batch", "1"]):
        out = str(tmp_path / ("out" + "_".join(extra)))
        r = _run([tp, pp, "--options", op, "-w", "400", "-h", "400", "-o", out] + extra)
        assert r.returncode == 0, r.stderr
        assert re.search(r"FPS:    [0-9.]+", r.stdout)
        got = np.array(Image.open(os.path.join(out, "r_0.png")))
        assert got.shape == (400, 400, 4) and np.array_equal(got, orc.rgba8(rgba)), extra
