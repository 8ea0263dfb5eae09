#!/usr/bin/env python3
"""Condenses a `rocprofv3 --kernel-trace --stats --output-format csv` directory into a short
per-kernel table for the timed steps of bench.py (the last `steps` x 16 aggregation launches
mark the timed region).  Usage: summarize.py <rocprof_dir> <steps> [aggregations_per_step [last_step_sequence.txt]] > summary.txt"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r'^void ', '', n)
    m = re.match(r'([\w:]+)', n)
    s = m.group(1) if m else n[:40]
    if s.startswith('Cijk'):
        s = 'hipBLASLt_GEMM_' + n[5:14] + '_MT' + re.search(r'MT(\w+?)_', n).group(1)
    if 'elementwise' in n:
        mm = re.search(r'(\w+Functor\w*|\w+_kernel_impl|launch_\w+)', n[n.find('<'):])
        s += ':' + (mm.group(1) if mm else '')
    return s[:78]


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    per_step_aggs = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    # bench.py with HIP graphs runs an eager pass over the same K steps after the timed region
    # (per-launch event timing): SKIP_TAIL_STEPS=K leaves it out, so that the table describes
    # the graph-replayed, timed steps
    import os
    skip = int(os.environ.get("SKIP_TAIL_STEPS", "0"))
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    # (the spin kernel that bench.py puts in front of its event-timed eager pass is no step work)
    rows = [r for r in rows if 'spin_kernel' not in r['Kernel_Name']]
    main_kernels = ('spmm_rows_kernel', 'spmm_plan_kernel', 'spmm_narrow_kernel')
    idx = [i for i, r in enumerate(rows) if any(k in r['Kernel_Name'] for k in main_kernels)]
    first = idx[-(steps + skip) * per_step_aggs]
    last = idx[-skip * per_step_aggs] if skip else len(rows)
    seg = rows[first:last]
    t0 = int(seg[0]['Start_Timestamp'])
    t1 = max(int(r['End_Timestamp']) for r in seg)
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    print(f"# timed region: {steps} steps, span {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, "
          f"{len(seg) / steps:.0f} kernels/step")
    agg = collections.defaultdict(lambda: [0, 0, None])
    for r in seg:
        k = short(r['Kernel_Name'])
        dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        agg[k][0] += 1
        agg[k][1] += dur
        agg[k][2] = (r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'])
    print(f"{'kernel':78s} {'n/step':>7s} {'ms/step':>8s} {'avg_us':>8s}  vgpr/sgpr/lds")
    for k, (c, dur, res) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"{k:78s} {c / steps:7.1f} {dur / steps / 1e6:8.3f} {dur / c / 1e3:8.1f}  {'/'.join(res)}")
    if len(sys.argv) > 4:   # kernel sequence of the last step, in launch order
        last = rows[idx[-(skip + 1) * per_step_aggs]:(idx[-skip * per_step_aggs] if skip else len(rows))]
        with open(sys.argv[4], 'w') as out:
            for r in last:
                out.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  "
                          f"{short(r['Kernel_Name'])}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}\n")


if __name__ == '__main__':
    main()
