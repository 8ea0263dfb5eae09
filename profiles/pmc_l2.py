#!/usr/bin/env python3
"""L2 hit rate of the aggregation launches from one rocprofv3 PMC pass of bench.py
(`--pmc TCC_HIT_sum TCC_MISS_sum`, csv output): totals over the aggregation kernels of the last
<steps> x <launches_per_step> launches.  Usage: pmc_l2.py <dir> <steps> <launches_per_step>"""
import csv
import glob
import json
import sys

MAIN = ("spmm_rows_kernel", "spmm_plan_kernel", "spmm_narrow_kernel")


def main():
    d, steps, per_step = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    for counter in ("TCC_HIT_sum", "TCC_MISS_sum"):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter
                and any(k in r["Kernel_Name"] for k in MAIN)]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        out[counter] = sum(float(r["Counter_Value"]) for r in rows[-steps * per_step:]) / (steps * per_step)
    out["l2_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
