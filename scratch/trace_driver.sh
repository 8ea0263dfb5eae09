#!/bin/bash
# Runs on the GPU box: kernel trace of one scratch/train_config.py configuration, reduced to the
# busy time per step interval and the kernel mix of the fullest one.  Usage: trace_driver.sh <config> [lines]
cfg=$1; lines=${2:-22}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/drv_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/drv_trace -- python3 scratch/train_config.py $cfg > gpurun_out/drv_$cfg.log 2>&1 || { tail -5 gpurun_out/drv_$cfg.log; exit 1; }
grep "train_seconds" gpurun_out/drv_$cfg.log | tail -2
python3 scratch/step_intervals.py gpurun_out/drv_trace $lines | tail -$((lines + 2))
rm -rf gpurun_out/drv_trace
