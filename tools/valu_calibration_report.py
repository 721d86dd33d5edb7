"""profiles/r3_valu_calibration.json from one run of tools/calibrate_valu2.sh:
tools/valu_calibration_report.py gpurun_out/r3b profiles/r3_valu_calibration.json
Per opcode: wave-level instructions per shader clock per SIMD at 2 and at 8 waves per SIMD, from the COUNTERS of the probe
launches themselves (SQ_INSTS_VALU / SQ_INSTS_SALU over SQ_BUSY_CYCLES / 32 shader engines) -- not from the probe's own
stamps: VALU issue is arbitrated oldest-first, so with more resident waves than saturate a SIMD the younger ones
starve before their first stamp and a wave's own lifetime says nothing about the rate."""
import json
import sys


def main():
    pre, out = sys.argv[1], sys.argv[2]
    d = json.load(open(pre + "_valu_probe_pmc.json"))
    names = [l.split("  wps")[0].strip() for l in open(pre + "_valu_probe.txt") if " wps 1:" in l]
    rows = sorted(d, key=lambda e: e["dispatch"])
    meas = [e for i, e in enumerate(rows) if i % 2 == 1]  # (warm-up, measured) pairs, kinds x wps (2, 8)
    table = []
    i = 0
    for nm in names:
        e = {"kind": nm}
        for wps in (2, 8):
            c = meas[i]["counters"]
            i += 1
            clk = c["SQ_BUSY_CYCLES"] / 32.0
            e["wps%d" % wps] = {"valu_per_clk_per_simd": c["SQ_INSTS_VALU"] / 1024.0 / clk,
                                "salu_per_clk_per_simd": c["SQ_INSTS_SALU"] / 1024.0 / clk,
                                "shader_clocks": clk, "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_INSTS_SALU": c["SQ_INSTS_SALU"],
                                "GRBM_GUI_ACTIVE_per_xcd": c["GRBM_GUI_ACTIVE"] / 8.0, "SQ_WAVES": c["SQ_WAVES"]}
        table.append(e)
    full = [e["kind"] for e in table if e["wps8"]["valu_per_clk_per_simd"] > 0.38]
    half = [e["kind"] for e in table if 0.2 < e["wps8"]["valu_per_clk_per_simd"] <= 0.38]
    quarter = [e["kind"] for e in table if 0.05 < e["wps8"]["valu_per_clk_per_simd"] <= 0.2]
    doc = {"what": __doc__, "classes": {"full_rate_0.41_to_0.45": full, "half_rate_0.235_to_0.29": half, "quarter_rate_0.12": quarter},
           "table": table}
    # the traversal kernel at a TRUE occupancy of k workgroups per CU (tuning key blocks_per_cu), real loads
    occ = {}
    for k in (1, 2, 4, 6):
        try:
            v = {c: x["mean"] for c, x in json.load(open("%s_real_pmc_k%d.json" % (pre, k)))["kernels"]["render_persist"].items()}
        except Exception:
            continue
        clk = v["SQ_BUSY_CYCLES"] / 32.0
        occ[str(k)] = {"shader_clocks": clk, "valu_per_clk_per_simd": v["SQ_INSTS_VALU"] / 1024.0 / clk,
                       "salu_per_clk_per_simd": v["SQ_INSTS_SALU"] / 1024.0 / clk,
                       "wait_any_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], "wait_inst_any_frac": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                       "active_inst_any_frac": v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"]}
    doc["render_persist_by_workgroups_per_cu"] = occ
    try:
        doc["render_persist_ms_by_workgroups_per_cu"] = [l.strip() for l in open(pre + "_real_by_occupancy.txt") if l.startswith("round 2")]
    except Exception:
        pass
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["classes"], indent=1))
    print(json.dumps(occ, indent=1))


if __name__ == "__main__":
    main()
