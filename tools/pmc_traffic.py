"""Builds profiles/pmc_traffic.json -- what bench.py's roofline block reads -- from the per-configuration
counter summaries of tools/profile_round.sh and the gather ceilings of tools/probe_ceiling.py.
usage: pmc_traffic.py OUT.json probe_ceiling.json c2=summary.json [c5=... c4=...]"""
import json
import os
import sys


def ceiling(probe, table_prefix, dependent=1):
    for g in probe["gather"]:
        if g["table"].startswith(table_prefix) and g["dependent"] == dependent and g["blocked"] == 0 and g["lines_per_gather"] == 21 \
                and g["waves_per_simd"] == 6:
            return g["line_accesses_per_clk_per_cu"]
    raise KeyError(table_prefix)


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
                      "6 waves per SIMD, table resident in L1 / L2 / beyond L2" % os.path.basename(probe_path)}
    for spec in sys.argv[3:]:
        wid, path = spec.split("=")
        k = json.load(open(path))["kernels"].get("render_persist")
        if not k:
            continue
        m = lambda c: k[c]["mean"] if c in k else None
        e = {"frames_per_launch": int(os.environ.get("RTO_FRAMES_PER_LAUNCH", "32")), "cus": 256,
             "fetch_bytes": m("FETCH_SIZE") * 1024.0, "write_bytes": m("WRITE_SIZE") * 1024.0,
             "tcp_line_accesses": m("TCP_TOTAL_CACHE_ACCESSES_sum"), "tcp_tcc_read_req": m("TCP_TCC_READ_REQ_sum"),
             "tcc_hit": m("TCC_HIT_sum"), "tcc_miss": m("TCC_MISS_sum"),
             # wave-level VALU instructions (a SIMD issues one per 4 clocks) and the quad-cycles the VALU was busy
             "valu_insts": m("SQ_INSTS_VALU"), "valu_active_quads": m("SQ_ACTIVE_INST_VALU"),
             # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md "DVFS give-back")
             "kernel_clocks": (m("GRBM_GUI_ACTIVE") or 0) / 8.0,
             "source": "%s: rocprofv3 --pmc, one pass per counter group (FETCH_SIZE; WRITE_SIZE; TCC/TCP; GRBM), "
                       "FETCH_SIZE as reported: 64 B per missed line for this kernel's scattered dword / 8-byte loads "
                       "(calibration: tools/pmc_probe.py)" % os.path.basename(path)}
        doc["workloads"][wid] = e
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
