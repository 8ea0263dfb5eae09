"""Ordered kernel list of ONE timed step out of a rocprofv3 kernel trace of bench.py.
Usage: python3 scratch/step_sequence.py <trace dir> <kernels per step> [skip tail steps]"""
import csv
import glob
import re
import sys

d, per = sys.argv[1], int(sys.argv[2])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 11
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# steps end with the optimizer's kernel (+ its step-counter increment)
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"] or "FusedAdam" in r["Kernel_Name"]]
k = len(ends) - skip - 3          # a step of the timed region (the eager pass comes after it)
step = rows[ends[k - 1] + 1:ends[k] + 1]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
for r in step:
    n = r["Kernel_Name"]
    n = re.sub(r"Cijk_(\w+?)_S_B.*?(MT\d+x\d+x\d+).*", r"GEMM \1 \2", n)
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::|std::array<char\*, \dul> ?", "", n)
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  +{(s - prev_end) / 1e3:5.1f}  {(e - s) / 1e3:7.1f} us  "
          f"grid {r['Grid_Size_X']:>9}  {n[:150]}")
    prev_end = e
