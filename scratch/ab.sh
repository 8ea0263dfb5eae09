#!/bin/bash
# A/B of the two aggregation kernels on the probe batch, with kernel trace + fabric counters.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0 1; do INCAGG_SPMM_SLICED=$m timeout -k 10 100 python3 scratch/spmm_probe.py 2>&1 | grep -E "spmm pair" | sed "s/^/sliced=$m /"; done
rm -rf gpurun_out/ab_*
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_trace -- python3 scratch/spmm_probe.py > /dev/null 2>&1
timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/ab_fetch -- python3 scratch/spmm_probe.py > /dev/null 2>&1
timeout -s KILL 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/ab_tcc -- python3 scratch/spmm_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ab_trace/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'incagg::spmm' in r['Name']: print(r['Name'].split('::')[1][:40], round(float(r['AverageNs'])/1e3,1), 'us')
for d in ('ab_fetch','ab_tcc'):
    f = glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'incagg::spmm' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('::')[1].split('<')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k, round(sum(v[3:])/len(v[3:])))
PY
