"""VALU issue-rate calibration (VERDICT r2 task 1a): tools/probe_valu.py [out.json] [--kinds 0,14] [--wps 1,2,4,6,8]
Each kind is ONE asm block of exactly 32 wave-level VALU instructions (csrc/probe_kernels.hip; the count is checked
against the disassembly by tests/test_codegen.py) run `iters` times by `wps` waves per SIMD on every CU.
Two rates per row, both in wave-level instructions per shader clock per SIMD:
  chip     = all instructions / (ticks from the first wave's start to the last wave's end x SIMDs)   <- the rate to quote
  per_wave = instructions of one wave x waves per SIMD / mean s_memtime ticks of a wave: NOT a rate once more waves are
             resident than saturate the SIMD -- issue is arbitrated oldest-first, younger waves starve before their first
             stamp and then run alone, so a wave's own lifetime stays short while the launch takes proportionally longer
`rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 tools/probe_valu.py --kinds 0 --wps 2` cross-checks both the count
and the clocks with the counters (tools/calibrate_valu.sh)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rt_octree_amd as R  # noqa: E402


def main():
    args = sys.argv[1:]
    kinds = wps_list = None
    iters = 20000
    out_path = None
    i = 0
    while i < len(args):
        if args[i] == "--kinds":
            kinds = [int(x) for x in args[i + 1].split(",")]
            i += 2
        elif args[i] == "--wps":
            wps_list = [int(x) for x in args[i + 1].split(",")]
            i += 2
        elif args[i] == "--iters":
            iters = int(args[i + 1])
            i += 2
        else:
            out_path = args[i]
            i += 1
    L = R.lib()
    names = []
    while L.rto_probe_valu_name(len(names)):
        names.append(L.rto_probe_valu_name(len(names)).decode())
    kinds = kinds if kinds is not None else list(range(len(names)))
    wps_list = wps_list or [1, 2, 4, 6, 8]
    out = (C.c_double * 6)()
    rows = []
    for k in kinds:
        for wps in wps_list:
            rc = L.rto_probe_valu(k, wps, iters, out)
            if rc != 0:
                raise SystemExit("rto_probe_valu failed rc %d" % rc)
            ms, cyc, waves, instr, span, cus = (out[j] for j in range(6))
            simds = cus * 4
            rows.append({"kind": names[k], "waves_per_simd": wps, "iters": iters, "ms": ms, "valu_insts_per_wave": instr,
                         "valu_insts_total": instr * waves, "ticks_per_wave": cyc, "span_ticks": span,
                         "ticks_per_ms": span / ms,
                         "clk_per_inst_one_wave_view": cyc / instr if instr else None,
                         "clk_per_block_one_wave_view": cyc / iters,
                         "insts_per_clk_per_simd_per_wave": instr * wps / cyc,
                         "insts_per_clk_per_simd_chip": instr * waves / (span * simds)})
            r = rows[-1]
            if instr > 0:
                print("%-40s wps %d: %.3f VALU instr/clk/SIMD (chip: all instructions / span x SIMDs)  %.2f clk per instr inside "
                      "one wave  %.2f ms  %.2f GHz" % (r["kind"], wps, r["insts_per_clk_per_simd_chip"],
                                                       r["clk_per_inst_one_wave_view"], ms, r["ticks_per_ms"] / 1e6), flush=True)
            else:  # a kind without VALU instructions: clocks per block of 32
                print("%-40s wps %d: %.1f clk per 32-instruction block inside one wave, span %.0f clk  %.2f ms" % (
                    r["kind"], wps, cyc / iters, span, ms), flush=True)
    if out_path:
        json.dump({"valu": rows, "note": __doc__}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
