"""Summarise rocprofv3 --pmc passes (csv output) into one JSON: mean counter value per dispatch for
each kernel family.  usage: pmc_summarize.py OUT.json DIR [DIR ...] [--traffic pmc_traffic.json]
Each DIR is the -d directory of one pass (`rocprofv3 --pmc A B -d DIR --output-format csv -- python3 bench.py ...`)."""
import csv
import glob
import json
import os
import sys

FAMILIES = ("render_persist", "shade_kernel", "raygen_kernel", "sample_kernel", "filter_fused", "filter_fast",
            "guidance_fused", "render_fast")


def family(name):
    for f in FAMILIES:
        if f in name:
            return f
    return None


def main():
    args = sys.argv[1:]
    traffic_out = None
    if "--traffic" in args:
        i = args.index("--traffic")
        traffic_out = args[i + 1]
        del args[i:i + 2]
    out, dirs = args[0], args[1:]
    acc = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    fam = family(row.get("Kernel_Name", ""))
                    if not fam:
                        continue
                    key = (fam, row["Counter_Name"])
                    # (pass directory, dispatch): a counter collected in two passes is averaged, not added up
                    disp = (d, row.get("Dispatch_Id") or row.get("Correlation_Id"))
                    acc.setdefault(key, {}).setdefault(disp, 0.0)
                    acc[key][disp] += float(row["Counter_Value"])
    kernels = {}
    for (fam, ctr), per in sorted(acc.items()):
        vals = list(per.values())
        kernels.setdefault(fam, {})[ctr] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
    doc = {"units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch, others raw counts per dispatch (summed over XCDs / SEs)",
           "kernels": kernels}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    rp = kernels.get("render_persist", {})
    if traffic_out and "FETCH_SIZE" in rp and "WRITE_SIZE" in rp:
        hbm = (rp["FETCH_SIZE"]["mean"] + rp["WRITE_SIZE"]["mean"]) * 1024.0
        with open(traffic_out, "w") as f:
            json.dump({"hbm_bytes_per_launch": hbm, "frames_per_launch": int(os.environ.get("RTO_FRAMES_PER_LAUNCH", "32")),
                       "source": os.path.basename(out) + " (render_persist<6>, FETCH_SIZE + WRITE_SIZE in "
                       "separate --pmc passes; the traversal's scattered dword loads count 64 B per touched line, as the "
                       "calibration probe tools/pmc_probe.py showed)"}, f, indent=1)
    print(json.dumps({k: {c: round(v["mean"], 1) for c, v in cs.items()} for k, cs in kernels.items()}, indent=1))


if __name__ == "__main__":
    main()
