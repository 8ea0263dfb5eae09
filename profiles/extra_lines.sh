#!/bin/bash
# Runs on the GPU box: bench lines of the other BASELINE.json configurations and of the
# incremental-aggregation (VR) step, one JSON line each under gpurun_out/<tag>_<name>.json.
# Usage: bash profiles/extra_lines.sh <tag>
tag=${1:-run}
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err || { tail -5 gpurun_out/${tag}_${name}.err; exit 1; }; python3 -c "
import json,sys; d=json.load(open('gpurun_out/${tag}_${name}.json')); r=d.get('roofline') or {}
print('${name}', round(d['value']/1e9,3),'G edges/s', round(d['ms_per_step'],3),'ms/step, roofline frac', round(r.get('frac',0),3), 'launch us', round(1e3*r.get('avg_launch_ms',0),1), 'cpu', round(d['cpu_baseline']['value']/1e9,4) if 'cpu_baseline' in d else None)"; }
run products_gcn2_vr --mode vr --cpu-seconds 4 && \
run arxiv_gcn --config arxiv-gcn --cpu-seconds 4 && \
run arxiv_gcn_vr --config arxiv-gcn --mode vr --cpu-seconds 4 && \
run arxiv_gcn_bn --config arxiv-gcn --batch-norm --cpu-seconds 4 && \
run arxiv_gcn_bn_vr --config arxiv-gcn --batch-norm --mode vr --cpu-seconds 4 && \
run reddit_appnp --config reddit-appnp --cpu-seconds 4 && \
run reddit_appnp_vr --config reddit-appnp --mode vr --cpu-seconds 4 && \
run products_sage_host --config products-sage-host --cpu-seconds 4 && \
run products_sage_hbm --config products-sage-host --tables hbm --cpu-seconds 4
