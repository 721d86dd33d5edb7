"""Code-generation guard for the traversal kernel (no GPU needed: hipcc cross-compiles gfx950).

render_persist issues two gathers per node visit -- 4 bytes of the traversal image for the lanes inside the tree, 8
bytes of the top grid for the others -- and its speed depends on both being in flight together.  Round 2 found a build
in which the register allocator had put an `s_waitcnt vmcnt(0)` BETWEEN the two loads (a temporary of the second branch
reused the first load's destination register): same instructions, same memory traffic, 14 % slower (DESIGN_HISTORY.md, "A
register-allocation cliff").  The kernel now forms both addresses before either load; this test keeps it that way."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rt-octree_amd", "csrc")
_ASM = {}
_REMARKS = {}


def makefile_flags(src):
    """the per-source flags of csrc/Makefile (`$(OBJ)/<name>.o: HIPFLAGS += ...`): what the shipped library is built with"""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^\$\(OBJ\)/%s\.o: HIPFLAGS \+= (.*)$" % re.escape(os.path.splitext(src)[0]), mk, re.M)
    return m.group(1).split() if m else []


def device_asm(src, extra=()):
    """gfx950 assembly of one device source, compiled once per test session with the Makefile's flags (+ extra); the same
    compilation's resource-usage remarks are kept for kernel_resources()"""
    extra = tuple(f for f in makefile_flags(src) if f not in extra) + tuple(extra)
    key = (src, tuple(extra))
    if key not in _ASM:
        import tempfile
        out = os.path.join(tempfile.mkdtemp(prefix="rto_codegen_"), src + ".s")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math"] + list(extra) +
                           ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                            os.path.join(CSRC, src), "-o", out], check=True, capture_output=True, text=True, timeout=900)
        with open(out) as f:
            _ASM[key] = f.read()
        _REMARKS[key] = r.stderr
        shutil.rmtree(os.path.dirname(out), ignore_errors=True)
    return _ASM[key]


def kernel_resources(src, extra=()):
    """{mangled kernel name: {"vgprs", "scratch", "occupancy"}} of one device source (the compiler's own remarks)"""
    device_asm(src, extra)
    extra = tuple(f for f in makefile_flags(src) if f not in extra) + tuple(extra)
    t = _REMARKS[(src, tuple(extra))]
    names = re.findall(r"Function Name: (\S+)", t)
    vg = [int(v) for v in re.findall(r" VGPRs: (\d+)", t)]
    sc = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", t)]
    oc = [int(v) for v in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", t)]
    assert len(names) == len(vg) == len(sc) == len(oc) and len(names) > 20
    return {n: {"vgprs": v, "scratch": s, "occupancy": o} for n, v, s, o in zip(names, vg, sc, oc)}


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_the_two_gathers_of_a_node_visit_are_issued_back_to_back(tmp_path):
    """the ONE-level walk (trees without the two-level image): top-grid entry and slot word are issued back to back"""
    text = device_asm("render_kernels.hip")
    for spp in (1, 6):  # the benchmark's instantiations (C5, C2 / C4)
        m = re.search(r"^_ZN3rto14render_persistILi%dELi32ELi8ELb0ELi0EEE[^\n]*\n(.*?)s_endpgm" % spp, text, re.S | re.M)
        assert m, "render_persist<%d,32,8,false> not found in the assembly" % spp
        body = [ln.split(";")[0].strip() for ln in m.group(1).splitlines()]
        body = [ln for ln in body if ln and not ln.startswith(".") or ln.startswith(".LBB")]
        grid = max(i for i, ln in enumerate(body) if ln.startswith("global_load_dwordx2"))  # the top-grid entry
        node = max(i for i, ln in enumerate(body[:grid]) if ln.startswith("global_load_dword "))  # the slot word
        between = body[node + 1:grid]
        assert len(between) <= 8, "unexpected code between the two gathers: %s" % between
        assert not any("vmcnt" in ln for ln in between), "a wait separates the two gathers: %s" % between
        assert not any(ln.startswith("v_") for ln in between), "VALU work between the two gathers: %s" % between


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_the_two_level_walk_has_one_gather_per_node_visit(tmp_path):
    """round 4: grid cells and wide nodes live in one array and a node visit is one uniform dword load -- no 8-byte grid
    entry, no second address computation anywhere in the kernel"""
    text = device_asm("render_kernels.hip")
    for spp in (1, 6):
        for stack in (0, 1):  # ancestor stack in LDS rows / in two registers
            m = re.search(r"^_ZN3rto14render_persistILi%dELi32ELi8ELb1ELi%dEEE[^\n]*\n(.*?)s_endpgm" % (spp, stack), text, re.S | re.M)
            assert m, "render_persist<%d,32,8,true,%d> not found in the assembly" % (spp, stack)
            assert "global_load_dwordx2" not in m.group(1)


LDS_STACK_SCRATCH, LDS_STACK_SCRATCH_32 = 24, 60  # bytes per lane today (16-24 for SPP <= 16)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_the_traversal_kernels_private_segments_are_what_is_recorded_here():
    """render_persist (64 % of the render stage) spilled 9 VGPRs and nothing watched that number.  Round 5: those
    spills were launch constants the compiler had hoisted out of the kernel's loops (bbox +- 1e-6 in double, 0.5 W, the NDC
    factors; later the constant halves of packed operations the SLP vectoriser had formed in the ray set-up -- render_kernels.hip
    is built without that pass now); the default batched traversal kernel -- render_persist<SPP, 32, 8, true, 1>, 8 waves per
    SIMD -- needs 57-64 VGPRs and NO private segment up to SPP 16 (SPP 32: its 32-entry flush), nor do the one-level-image
    instantiation and the single-frame kernel at 5 waves (88 VGPRs).  The two-level form
    with its ancestor stack in LDS rows (trees deeper than four levels below the top grid) keeps 16-24 bytes, in the ray set-up:
    sizes recorded here so that a change of them is a decision, not an accident."""
    res = kernel_resources("render_kernels.hip")
    for spp in (1, 2, 3, 4, 6, 8, 16, 32):
        k = res["_ZN3rto14render_persistILi%dELi32ELi8ELb1ELi1EEEvNS_7TreeDevENS_6OptDevENS_10FrameBatchEPyPjj" % spp]
        assert k["occupancy"] == 8 and k["vgprs"] <= 64, (spp, k)
        assert k["scratch"] <= (56 if spp == 32 else 0), "render_persist<%d, wide>: %d bytes of scratch per lane" % (spp, k["scratch"])
        kl = res["_ZN3rto14render_persistILi%dELi32ELi8ELb1ELi0EEEvNS_7TreeDevENS_6OptDevENS_10FrameBatchEPyPjj" % spp]
        assert kl["occupancy"] == 8 and kl["scratch"] <= (LDS_STACK_SCRATCH_32 if spp == 32 else LDS_STACK_SCRATCH), (spp, kl)
        k1 = res["_ZN3rto14render_persistILi%dELi32ELi8ELb0ELi0EEEvNS_7TreeDevENS_6OptDevENS_10FrameBatchEPyPjj" % spp]
        assert k1["occupancy"] == 8 and k1["scratch"] <= (40 if spp == 32 else 0), (spp, k1)
    # the single-frame kernel on the two-level image: ancestor stack in LDS rows (0) / the register-stack restart (1)
    fast = [v for n, v in res.items() if n.startswith("_ZN3rto11render_fastILi6ELb0ELb1ELi")]
    assert len(fast) == 2 and all(f["occupancy"] == 5 and f["scratch"] == 0 for f in fast), fast


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_the_valu_probe_loops_hold_exactly_the_instructions_they_claim(tmp_path):
    """tools/probe_valu.py divides instruction counts by clocks: the count must be what the code object holds.  Every
    probe kind's loop is ONE asm block of 32 VALU instructions (0 for the SALU / LDS kinds) plus the three SALU
    instructions of the loop itself -- nothing the compiler added, packed or folded (round 2's C-level probe lost part of
    its nominal count that way)."""
    asm = tmp_path / "probe_kernels.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math",
                    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
                    os.path.join(CSRC, "probe_kernels.hip"), "-o", str(asm)], check=True, capture_output=True, timeout=900)
    text = asm.read_text()
    src = open(os.path.join(CSRC, "probe_kernels.hip")).read()
    table = re.findall(r'\{"([^"]+)", (kBlk|\d+), valu_probe_kernel<(\d+)>\}', src)
    assert len(table) >= 60
    for name, count, kind in table:
        want = 32 if count == "kBlk" else int(count)
        m = re.search(r"^_ZN12_GLOBAL__N_117valu_probe_kernelILi%sEEEviffPfPy:[^\n]*\n(.*?)s_endpgm" % kind, text, re.S | re.M)
        assert m, name
        body = [ln.split(";")[0].strip() for ln in m.group(1).splitlines()]
        body = [ln for ln in body if ln and (not ln.startswith(".") or ln.startswith(".LBB"))]
        loops = []
        for i, ln in enumerate(body):
            if ln.startswith("s_cbranch"):
                tgt = ln.split()[-1] + ":"
                if tgt in body[:i]:
                    loops.append(body[body.index(tgt) + 1:i + 1])
        loop = max(loops, key=len)
        valu = [ln for ln in loop if ln.startswith("v_")]
        assert len(valu) == want, (name, len(valu), want)
        other = [ln for ln in loop if not ln.startswith("v_")]
        assert len([ln for ln in other if ln.startswith(("s_add_i32", "s_cmp", "s_cbranch"))]) == 3, (name, other[-4:])


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_no_kernel_of_the_denoise_stage_uses_scratch(tmp_path):
    """Round 3 (tools/contention_determinism.py, profiles/r3_contention_determinism.txt): the GuidanceNet instantiations that
    spilled registers were the slowest MFMA kernels of the path -- and the company in which the bit-exact filter first lost
    its determinism on a shared GPU (see the next test for the cause).  The spills are gone (launch bounds per
    instantiation); this keeps them gone for every kernel of guidance_kernels.hip and filter_kernels.hip."""
    for src in ("guidance_kernels.hip", "filter_kernels.hip"):
        # (the flags the Makefile builds each source with: filter_kernels.hip without the SLP vectoriser, see the next test)
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math"] +
                           (["-fno-slp-vectorize"] if src == "filter_kernels.hip" else []) +
                           ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", "--cuda-device-only",
                            "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, src), "-o", str(tmp_path / (src + ".o"))],
                           check=True, capture_output=True, text=True, timeout=900)
        names = re.findall(r"Function Name: (\S+)", r.stderr)
        scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
        assert len(names) == len(scratch) and len(names) >= 4, (src, len(names), len(scratch))
        bad = [(n, s) for n, s in zip(names, scratch) if s != 0]
        assert not bad, "kernels of %s with a private segment: %s" % (src, bad)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_no_kernel_holds_v_pk_fma_f32(tmp_path):
    """Round 3 (tools/scratch_hazard_probe.py, profiles/r3_scratch_hazard_probe.txt): the bit-exact filter, the only kernel
    that held v_pk_fma_f32, returned wrong values in lanes 48..63 while another process ran MFMA-dense kernels on the same
    GPU -- it differed from its own repeat in 507 of 800 runs, and in 0 of 800 once it was built from scalar FMAs (same
    speed).  filter_kernels.hip is therefore compiled without the SLP vectoriser (which re-forms the packed FMAs);
    this test builds every device source the way the Makefile does and looks for the instruction."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert re.search(r"filter_kernels\.o: HIPFLAGS \+= -fno-slp-vectorize", mk), "the Makefile no longer disables SLP for filter_kernels.hip"
    for src, extra in (("filter_kernels.hip", ["-fno-slp-vectorize"]), ("guidance_kernels.hip", []), ("render_kernels.hip", [])):  # (probe_kernels.hip holds the instruction on purpose: the VALU calibration probe)
        text = device_asm(src, extra)
        assert "s_endpgm" in text
        assert "v_pk_fma_f32" not in text, "%s: %d v_pk_fma_f32" % (src, text.count("v_pk_fma_f32"))
