#!/bin/bash
# Runs on the GPU box: the default bench line, a rocprofv3 kernel trace of the same command and
# the two PMC passes for fabric traffic; leaves everything under gpurun_out/$1_*.
# Usage: bash profiles/collect.sh <tag>
tag=${1:-run}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/${tag}_trace gpurun_out/${tag}_fetch gpurun_out/${tag}_write
echo "bench line"; timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err || exit 1
echo "kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --sweep off --epoch-pass off > gpurun_out/${tag}_trace.log 2>&1 || exit 1
SKIP_TAIL_STEPS=11 python3 profiles/summarize.py gpurun_out/${tag}_trace 10 16 > gpurun_out/${tag}_kernel_summary.txt || exit 1
cp gpurun_out/${tag}_trace/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
echo "pmc fetch"; timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --sweep off --epoch-pass off > gpurun_out/${tag}_fetch.log 2>&1 || exit 1
echo "pmc write"; timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --sweep off --epoch-pass off > gpurun_out/${tag}_write.log 2>&1 || exit 1
python3 profiles/pmc_traffic.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write 10 16 > gpurun_out/${tag}_pmc_traffic.json || exit 1
echo "pmc l2"; timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_l2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --sweep off --epoch-pass off > gpurun_out/${tag}_l2.log 2>&1 || exit 1
python3 profiles/pmc_l2.py gpurun_out/${tag}_l2 10 16 > gpurun_out/${tag}_pmc_l2.json || exit 1
rm -rf gpurun_out/${tag}_l2
cat gpurun_out/${tag}_pmc_traffic.json
head -12 gpurun_out/${tag}_kernel_summary.txt
