#!/bin/bash
# round 3, run I: kernel statistics of the config-4 (U-ViT) leg, weight gradients on the main stream (true per-kernel durations)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -rf $O/prof_uvit
MUSE_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_uvit -o u -- python bench.py --uvit-leg 128,256,4 > $O/r3i_uvit.txt 2>&1
tail -1 $O/r3i_uvit.txt | cut -c1-200
f=$(find $O/prof_uvit -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/r3i_uvit_kernel_stats_serial.csv && head -32 "$f" | cut -c1-180
find $O/prof_uvit -name "*kernel_trace*" -delete
