#!/bin/bash
# rocprof kernel ranking of the serial bench on the final tree of round 2 (pairs with profiles/r02_final2_bench.json)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -rf $O/prof_final2
MUSE_WGRAD_STREAM=0 timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final2 -o r2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-prefetch > $O/r2f4_prof.txt 2>&1
f=$(find $O/prof_final2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2f4_kernel_stats.csv && head -8 "$f" | cut -c1-150
find $O/prof_final2 -name "*kernel_trace*" -delete
tail -c 600 $O/r2f4_prof.txt
