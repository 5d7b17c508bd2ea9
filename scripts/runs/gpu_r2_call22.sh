#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -rf $O/prof22
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof22 -o uvit -- python scripts/uvit_bench.py 64 3 bf16 256 adamw > $O/r2_call22_uvit.txt 2>&1
tail -4 $O/r2_call22_uvit.txt
f=$(find $O/prof22 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2_call22_uvit_kernel_stats.csv && head -45 "$f" | cut -c1-170
find $O/prof22 -name "*kernel_trace*" -size +8M -delete
