#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of one bench configuration, reduced to the
# per-step kernel mix of the timed steps (aggregation launches per step taken from the bench
# line's own count).  Usage: bash profiles/trace_config.sh <tag> <bench args...>
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/${tag}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --sweep off --epoch-pass off > gpurun_out/${tag}_trace.log 2>&1 || { tail -5 gpurun_out/${tag}_trace.log; exit 1; }
per_step=$(python3 -c "
import json,sys
for l in open('gpurun_out/${tag}_trace.log'):
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print(d['roofline']['launches']//d['steps']); break
") || exit 1
SKIP_TAIL_STEPS=11 python3 profiles/summarize.py gpurun_out/${tag}_trace 10 $per_step gpurun_out/${tag}_last_step.txt > gpurun_out/${tag}_kernel_summary.txt || exit 1
rm -rf gpurun_out/${tag}_trace
head -${TRACE_LINES:-26} gpurun_out/${tag}_kernel_summary.txt
