"""Per-kernel GPU time of scratch/build_trace.py from its rocprofv3 kernel trace (argv[1]: trace
directory; 123 builds per run)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = [i for i, n in enumerate(names) if "incagg" in n][0]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[first:]:
    n = r["Kernel_Name"]
    if "rocprim" in n:      # which primitive, which key / value types
        import re
        m = re.search(r"detail::(\w+)<", n[60:])
        n = "rocprim " + (m.group(1) if m else "?") + " | " + n[-110:]
    n = n[:150]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("kernels", len(rows) - first, "busy ms per build", tot / 123 / 1e6)
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{t / 123 / 1e3:8.1f} us/build {c / 123:6.1f}/build  {k}")
