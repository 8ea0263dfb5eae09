#!/bin/bash
# kernel trace + fabric counters of scratch/grouped_prof.py; env MODE etc. pass through
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pf_*
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf_trace -- python3 scratch/grouped_prof.py > /dev/null 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf_fetch -- python3 scratch/grouped_prof.py > /dev/null 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pf_tcc -- python3 scratch/grouped_prof.py > /dev/null 2>&1 || exit 1
timeout -s KILL 100 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pf_sq -- python3 scratch/grouped_prof.py > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pf_trace/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'incagg::spmm' in r['Name']: print(r['Name'].split('::')[1][:40], round(float(r['AverageNs'])/1e3,1), 'us')
for d in ('pf_fetch','pf_tcc','pf_sq'):
    f = glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'incagg::spmm' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('::')[1].split('<')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k, round(sum(v[3:])/len(v[3:])))
PY
