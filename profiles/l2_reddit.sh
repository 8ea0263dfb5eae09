#!/bin/bash
# Runs on the GPU box: L2 hit rate of the aggregation launches of the reddit-sized APPNP
# configuration with plans ordered by row length only and with the chunk-by-chunk order
# (one rocprofv3 PMC pass each).  Usage: bash profiles/l2_reddit.sh <tag>
tag=${1:-run}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for mode in length chunked; do
  if [ $mode = length ]; then export INCAGG_PLAN_LOCAL_ROWS=1073741824; else unset INCAGG_PLAN_LOCAL_ROWS; fi
  rm -rf gpurun_out/${tag}_l2_$mode
  echo "pmc l2 $mode"
  timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_l2_$mode -- python3 bench.py --config reddit-appnp --steps 3 --warmup 1 --no-cpu-baseline --sweep off --epoch-pass off > gpurun_out/${tag}_l2_$mode.log 2>&1 || exit 1
  python3 profiles/pmc_l2.py gpurun_out/${tag}_l2_$mode 2 20 > gpurun_out/${tag}_reddit_pmc_l2_$mode.json || exit 1
  rm -rf gpurun_out/${tag}_l2_$mode
  cat gpurun_out/${tag}_reddit_pmc_l2_$mode.json
done
