"""Per-dispatch counter table of rocprofv3 --pmc passes (csv): tools/pmc_dump.py OUT.json KERNEL_SUBSTRING DIR [DIR ...]
-> for every dispatch of a kernel whose name contains the substring, the counters summed over XCDs / SEs."""
import csv
import glob
import json
import os
import sys


def main():
    out, needle, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    rows = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "")
                    if needle not in name:
                        continue
                    disp = int(row.get("Dispatch_Id") or row.get("Correlation_Id"))
                    e = rows.setdefault((os.path.basename(d.rstrip("/")), disp), {"kernel": name[:120], "grid": row.get("Grid_Size"),
                                                                                 "counters": {}})
                    e["counters"][row["Counter_Name"]] = e["counters"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    doc = [{"pass": k[0], "dispatch": k[1], **v} for k, v in sorted(rows.items())]
    json.dump(doc, open(out, "w"), indent=1)
    for e in doc:
        print(e["pass"], e["dispatch"], e["kernel"][:60], e["grid"], {k: round(v) for k, v in e["counters"].items()})


if __name__ == "__main__":
    main()
