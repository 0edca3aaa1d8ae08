#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc passes written by tools/pmc_passes.sh.

    python tools/pmc_summary.py gpurun_out/pmc_r2 [--kernel c4_selfplay_split_kernel] [--json out.json]

For every counter: the mean per launch over the second half of the kernel's dispatches (steady state)."""
import argparse
import csv
import glob
import json
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--kernel", default="c4_selfplay_split_kernel")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = {}
    for f in sorted(glob.glob(os.path.join(a.out_dir, "*", "**", "*counter_collection.csv"), recursive=True)):
        rows = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if a.kernel not in r.get("Kernel_Name", ""):
                    continue
                rows.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                rows[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for name, per in rows.items():
            vals = [per[k] for k in sorted(per, key=lambda x: int(x))]
            half = vals[len(vals) // 2:]
            res[name] = dict(mean_per_launch=sum(half) / len(half), launches=len(vals), launches_averaged=len(half))
    for k in sorted(res):
        print("%-28s %18.1f   (%d of %d launches)" % (k, res[k]["mean_per_launch"], res[k]["launches_averaged"], res[k]["launches"]))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(res, fh, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
