"""Builds profiles/pmc_traffic.json -- what bench.py's roofline block reads -- from the per-configuration
counter summaries of tools/profile_round.sh and the gather ceilings of tools/probe_ceiling.py.
usage: pmc_traffic.py OUT.json probe_ceiling.json c2=summary.json[:bench_line.json] [c5=... c4=...]
Every workload entry carries the identity of the kernel code it was measured on (bench.kernel_code_id(): sha256 over the
traversal kernel's sources) and the frames per launch of the profiled run, read from the bench line written under
rocprofv3 (bench_line.json) -- bench.py refuses counters of other code."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_code_id only: no GPU work at import)


def ceiling(probe, table_prefix, dependent=1, waves=(6, 8)):
    """lines per clock per CU the probe sustained for 21-line gathers at the given waves per SIMD (mean over `waves`: the
    dependent figures are latency x concurrency, so they scale with the resident waves -- the traversal kernel holds 7)"""
    vals = []
    for w in waves:
        for g in probe["gather"]:
            if g["table"].startswith(table_prefix) and g["dependent"] == dependent and g["blocked"] == 0 and g["lines_per_gather"] == 21 \
                    and g["waves_per_simd"] == w:
                vals.append(g["line_accesses_per_clk_per_cu"])
                break
    if not vals:
        raise KeyError(table_prefix)
    return sum(vals) / len(vals)


def main():
    out, probe_path = sys.argv[1], sys.argv[2]
    doc = {"units": "bytes / raw counts per render_persist launch (means over the profiled launches)", "workloads": {}}
    if os.path.exists(probe_path):
        probe = json.load(open(probe_path))
        doc["ceilings"] = {
            "l1_hit_lines_per_clk": ceiling(probe, "16 KiB"), "l2_lines_per_clk": ceiling(probe, "2 MiB"),
            "mall_lines_per_clk": ceiling(probe, "64 MiB"),
            # the same with FOUR independent gathers in flight per wave: what the L1 sustains when nothing waits on a
            # previous load -- an upper bound on its rates, hence a lower bound on the share of the kernel it needs
            "independent": {"l1_hit_lines_per_clk": ceiling(probe, "16 KiB", 0), "l2_lines_per_clk": ceiling(probe, "2 MiB", 0),
                            "mall_lines_per_clk": ceiling(probe, "64 MiB", 0)},
            "source": "tools/probe_ceiling.py (%s): one dependent dword gather per wave touching 21 distinct 64-B lines, "
                      "mean of 6 and 8 waves per SIMD (the traversal kernel holds 7), table resident in L1 / L2 / beyond L2" % os.path.basename(probe_path)}
    cal = os.path.join(ROOT, "profiles", "r3_valu_calibration.json")
    if os.path.exists(cal):
        t = {e["kind"]: e for e in json.load(open(cal))["table"]}
        doc["valu_ceiling"] = {
            "traversal_mix_insts_per_clk_per_simd": t["traversal mix (16 opcodes)"]["wps8"]["valu_per_clk_per_simd"],
            "full_rate_insts_per_clk_per_simd": t["v_fma_f32"]["wps8"]["valu_per_clk_per_simd"],
            "half_rate_insts_per_clk_per_simd": t["v_med3_f32"]["wps8"]["valu_per_clk_per_simd"],
            "source": "profiles/r3_valu_calibration.json: asm probe, 8 waves per SIMD, SQ_INSTS_VALU / (SQ_BUSY_CYCLES / 32) / 1024 SIMDs"}
        # the traversal loop itself with both gathers stubbed (-DRTO_STUB_LOADS): what ITS instruction stream sustains with no
        # memory in the way; the ceiling quoted is the larger of the two
        import glob
        stubs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_stubbed_loads_pmc.json")))  # (the latest bundle's)
        stub = stubs[-1] if stubs else ""
        vc = doc["valu_ceiling"]
        vc["ceiling_insts_per_clk_per_simd"] = vc["traversal_mix_insts_per_clk_per_simd"]
        if stub and os.path.exists(stub):
            k = json.load(open(stub))["kernels"].get("render_persist", {})
            if "SQ_INSTS_VALU" in k and ("SQ_BUSY_CYCLES" in k or "GRBM_GUI_ACTIVE" in k):
                clk = k["SQ_BUSY_CYCLES"]["mean"] / 32.0 if "SQ_BUSY_CYCLES" in k else k["GRBM_GUI_ACTIVE"]["mean"] / 8.0
                vc["stubbed_loop_insts_per_clk_per_simd"] = k["SQ_INSTS_VALU"]["mean"] / 1024.0 / clk
                vc["stubbed_loop_source"] = os.path.relpath(stub, ROOT)
                vc["ceiling_insts_per_clk_per_simd"] = max(vc["ceiling_insts_per_clk_per_simd"], vc["stubbed_loop_insts_per_clk_per_simd"])
    code_id = bench.kernel_code_id()
    for spec in sys.argv[3:]:
        wid, path = spec.split("=")
        line = None
        if ":" in path:
            path, line = path.split(":")
        k = json.load(open(path))["kernels"].get("render_persist")
        if not k:
            continue
        fpl = None
        if line and os.path.exists(line):
            try:
                fpl = json.loads([l for l in open(line) if l.startswith("{")][-1])["config"]["frames_per_launch"]
            except Exception:
                fpl = None
        if fpl is None:
            fpl = float(os.environ.get("RTO_FRAMES_PER_LAUNCH", "100"))
        m = lambda c: k[c]["mean"] if c in k else None
        wc = m("SQ_WAVE_CYCLES")
        e = {"frames_per_launch": int(round(fpl)), "cus": 256, "kernel_code_id": code_id,
             "salu_insts": m("SQ_INSTS_SALU"),
             "wait_any_frac": (m("SQ_WAIT_ANY") / wc) if wc and m("SQ_WAIT_ANY") else None,
             "wait_inst_any_frac": (m("SQ_WAIT_INST_ANY") / wc) if wc and m("SQ_WAIT_INST_ANY") else None,
             "active_inst_any_frac": (m("SQ_ACTIVE_INST_ANY") / wc) if wc and m("SQ_ACTIVE_INST_ANY") else None,
             "lanes_per_valu_inst": (m("SQ_THREAD_CYCLES_VALU") / m("SQ_INSTS_VALU")) if m("SQ_THREAD_CYCLES_VALU") and m("SQ_INSTS_VALU") else None,
             "fetch_bytes": m("FETCH_SIZE") * 1024.0, "write_bytes": m("WRITE_SIZE") * 1024.0,
             "tcp_line_accesses": m("TCP_TOTAL_CACHE_ACCESSES_sum"), "tcp_tcc_read_req": m("TCP_TCC_READ_REQ_sum"),
             "tcc_hit": m("TCC_HIT_sum"), "tcc_miss": m("TCC_MISS_sum"),
             # wave-level VALU instructions and the quad-cycles the VALU was busy
             "valu_insts": m("SQ_INSTS_VALU"), "valu_active_quads": m("SQ_ACTIVE_INST_VALU"),
             # shader clocks of the launch: SQ_BUSY_CYCLES is summed over the 32 shader engines (GRBM_GUI_ACTIVE / 8 XCDs agrees
             # for long kernels and overstates short ones)
             "kernel_clocks": (m("SQ_BUSY_CYCLES") / 32.0) if m("SQ_BUSY_CYCLES") else (m("GRBM_GUI_ACTIVE") or 0) / 8.0,
             "source": "%s: rocprofv3 --pmc, one pass per counter group (FETCH_SIZE; WRITE_SIZE; TCC/TCP; GRBM), "
                       "FETCH_SIZE as reported: 64 B per missed line for this kernel's scattered dword / 8-byte loads "
                       "(calibration: tools/pmc_probe.py)" % os.path.basename(path)}
        doc["workloads"][wid] = e
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
