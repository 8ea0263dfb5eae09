"""From a rocprofv3 kernel trace of a training-driver run: GPU busy time and span between
consecutive optimizer launches (= one training step with everything around it), and the kernel
mix of the last complete interval.  argv[1]: trace directory."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"] or "multi_tensor_apply" in r["Kernel_Name"]]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for a, b in list(zip(opt[:-1], opt[1:]))[-6:]:
    seg = rows[a + 1:b + 1]
    print(f"interval of {len(seg):4d} kernels: busy {sum(map(dur, seg)) / 1e6:7.3f} ms, span "
          f"{(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6:7.3f} ms")
a, b = max(list(zip(opt[:-1], opt[1:]))[-8:], key=lambda ab: ab[1] - ab[0])   # the fullest recent interval
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[a + 1:b + 1]:
    k = r["Kernel_Name"][:100]
    agg[k][0] += 1
    agg[k][1] += dur(r)
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{t / 1e3:9.1f} us {c:5d} x  {k}")
