#!/bin/bash
# round 3, run J: stream / queue placement and family overlap of the final plain step (kernel trace through scripts/overlap_report.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -rf $O/prof_j
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_j -o t -- python scripts/exp/host_bound.py full > $O/r3j_run.txt 2>&1
grep "host enqueue" $O/r3j_run.txt
f=$(find $O/prof_j -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/overlap_report.py "$f" --last-ms 400 > $O/r3j_overlap.txt 2>&1 && head -40 $O/r3j_overlap.txt | cut -c1-150
find $O/prof_j -name "*kernel_trace*" -delete
