"""Ceilings for the traversal kernel, measured on the box (tools/probe_ceiling.py [out.json]):
gather sweep (distinct 64-B lines per wave-level dword gather x table size x dependent/independent) and
VALU issue rate.  Prints one JSON document; the judged copy lives in profiles/."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rt_octree_amd as R  # noqa: E402

L = R.lib()
CUS = 256
res = {"gather": [], "valu": []}
out = (C.c_double * 4)()
for table_mb, label in ((0.015625, "16 KiB (L1-resident)"), (2, "2 MiB (L2-resident: the top grid)"),
                        (16, "16 MiB"), (64, "64 MiB (MALL-resident: the traversal image)")):
    for dep in (1, 0):
        for blocked in (0, 1):
            for K in (1, 2, 4, 8, 16, 21, 32, 64):
                for wps in ((6,) if not (K == 21 and blocked == 0) else (2, 4, 6, 8)):
                    iters = 2048 if table_mb < 1 else 1024
                    rc = L.rto_probe_gather_sweep(int(table_mb * (1 << 20)), K, blocked, dep, wps, iters, 3, out)
                    if rc != 0:
                        raise SystemExit("probe failed rc %d" % rc)
                    ms, cyc, waves, gathers = out[0], out[1], out[2], out[3]
                    waves_per_cu = waves / CUS
                    # per CU: gathers issued by its waves over the mean wave lifetime in shader clocks
                    lines_per_clk_cu = K * gathers * waves_per_cu / cyc
                    res["gather"].append({
                        "table": label, "dependent": dep, "blocked": blocked, "lines_per_gather": K, "waves_per_simd": wps,
                        "ms": ms, "cycles_per_wave": cyc, "clock_ghz": cyc / (ms * 1e6),
                        "gathers_per_us_chip": gathers * waves / (ms * 1e3),
                        "line_accesses_per_clk_per_cu": lines_per_clk_cu,
                        "lane_loads_per_clk_per_cu": 64 * gathers * waves_per_cu / cyc})
for kind, name in ((0, "v_fma_f32"), (1, "integer xor/add/bfe"), (2, "traversal mix")):
    for wps in (1, 2, 4, 6, 8):
        rc = L.rto_probe_valu(kind, wps, 4096, out)
        if rc != 0:
            raise SystemExit("valu probe failed rc %d" % rc)
        ms, cyc, waves, instr = out[0], out[1], out[2], out[3]
        res["valu"].append({"kind": name, "waves_per_simd": wps, "ms": ms, "cycles_per_wave": cyc,
                            "clock_ghz": cyc / (ms * 1e6),
                            "nominal_instr_per_clk_per_simd": instr * wps / cyc})
txt = json.dumps(res, indent=1)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt)
for g in res["gather"]:
    print("%-44s dep %d blk %d K %2d wps %d: %.3f lines/clk/CU, %.2f lane-loads/clk/CU, %.2f GHz" % (
        g["table"], g["dependent"], g["blocked"], g["lines_per_gather"], g["waves_per_simd"],
        g["line_accesses_per_clk_per_cu"], g["lane_loads_per_clk_per_cu"], g["clock_ghz"]))
for v in res["valu"]:
    print("valu %-22s wps %d: %.3f instr/clk/SIMD (nominal), %.2f GHz" % (v["kind"], v["waves_per_simd"], v["nominal_instr_per_clk_per_simd"], v["clock_ghz"]))
