"""From a rocprofv3 kernel trace of `bench.py --sweep on --steps 2 --warmup 1 --no-cpu-baseline
--epoch-pass off`: GPU busy time, span and kernel mix of the LAST refresh sweep of the run (the
timed replay; bench.py issues it after its steps).  A sweep is recognised by its aggregation
launches that come in a row without an optimizer kernel in between.  argv[1]: trace directory."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
opt = [i for i, n in enumerate(names) if "adam_bump_kernel" in n or "FusedOptimizer" in n]
# the last stretch without optimizer kernels that holds more than 100 aggregation launches
cuts = [-1] + opt + [len(rows)]
best = None
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = rows[a + 1:b]
    if sum("spmm_" in r["Kernel_Name"] for r in seg) > 100:
        best = seg
sw = best
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("sweep kernels", len(sw), "busy ms", sum(map(dur, sw)) / 1e6, "span ms",
      (int(sw[-1]["End_Timestamp"]) - int(sw[0]["Start_Timestamp"])) / 1e6)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sw:
    n = r["Kernel_Name"]
    m = re.search(r"(MT\d+x\d+x\d+)", n)
    e = re.search(r"(direct_copy|addcmul|threshold|launch_clamp|FillFunctor|CatArray|CUDAFunctor_add|MulFunctor|"
                  r"AUnaryFunctor|BinaryFunctor<float, float, float, [A-Za-z_:]*|index_elementwise|vectorized_gather)", n)
    k = ("GEMM " + m.group(1)) if m else (e.group(1) + " grid " + r["Grid_Size_X"]) if e else \
        re.sub(r"<.*", "", n.replace("void ", "").replace("at::native::", ""))[:50]
    agg[k][0] += 1
    agg[k][1] += dur(r) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{t / 1e3:8.2f} ms {c:6d} x {t / c:7.1f} us  {k}")
