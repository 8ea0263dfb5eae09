#!/usr/bin/env python3
"""Fabric traffic per aggregation launch from two rocprofv3 PMC passes of bench.py
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, one run each, csv output).  Both counters are in
KiB-sized units of 1024 B as rocprofv3 reports them for gfx950; FETCH_SIZE counts 128-byte
requests at 64 B and is doubled (checked against a pure row gather of known size, see
r01_pmc_notes.txt), WRITE_SIZE is exact.  A "launch" is one call of the aggregation entry
point: the aggregation kernel plus, when the plan holds rows cut into several segments, the
small launch that zeroes those rows first.  The output names the SHA-256 of the kernel source
it was measured with; bench.py reports the figure only while that still matches.
Usage: pmc_traffic.py <fetch_dir> <write_dir> <steps> <launches_per_step> > traffic.json"""
import csv
import glob
import hashlib
import json
import os
import sys

KERNEL_SOURCE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                             "incagg-gnn_amd", "csrc", "aggregate.hip")

MAIN = ("spmm_rows_kernel", "spmm_plan_kernel", "spmm_narrow_kernel")
SECOND = ("zero_rows_kernel",)


def per_launch(d, counter, steps, per_step):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    mains = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in MAIN)]
    # (with HIP graphs bench.py appends an eager pass over the same K steps; both passes run
    # the same launches, so the tail of the run is representative either way)
    first = mains[-steps * per_step]
    total = sum(float(r["Counter_Value"]) for r in rows[first:]
                if any(k in r["Kernel_Name"] for k in MAIN + SECOND))
    return total / (steps * per_step)


def main():
    fetch_dir, write_dir, steps, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    fetch = per_launch(fetch_dir, "FETCH_SIZE", steps, per_step)
    write = per_launch(write_dir, "WRITE_SIZE", steps, per_step)
    print(json.dumps({
        "workload": "bench.py default (gas, products, 150 parts, F=256), timed steps",
        "fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write,
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as is",
        "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
        "kernel_source_sha256": hashlib.sha256(open(KERNEL_SOURCE, "rb").read()).hexdigest(),
    }, indent=1))


if __name__ == "__main__":
    main()
