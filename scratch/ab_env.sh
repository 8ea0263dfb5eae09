#!/bin/bash
# A/B of environment switches over a bench line (steps only): one line per setting.
# Usage: BENCH_ARGS="--config arxiv-gcn" bash scratch/ab_env.sh <tag> "VAR=a VAR2=b" "VAR=c" ...
tag=$1; shift
out=gpurun_out/${tag}_ab.txt
: > $out
for setting in "$@"; do
  echo "== $setting" >> $out
  env $setting timeout -k 10 300 python3 bench.py $BENCH_ARGS --steps 30 --warmup 5 --no-cpu-baseline --sweep off --epoch-pass off 2>> gpurun_out/${tag}_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms_per_step', round(d['ms_per_step'],4), 'agg_avg_us', round(d['roofline']['avg_launch_ms']*1e3,2), 'frac', round(d['roofline']['frac'],4))" >> $out || { echo "run failed" >> $out; exit 1; }
done
cat $out
