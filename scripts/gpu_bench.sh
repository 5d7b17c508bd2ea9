#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python bench.py --steps ${STEPS:-8} --warmup 2 ${BENCH_ARGS} > $O/bench.txt 2>&1; echo "bench exit $?" >> $O/bench.txt; tail -3 $O/bench.txt | cut -c1-3500
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/prof.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -32 "$f" | cut -c1-160
find $O/prof -name "*kernel_trace*" -size +6M -delete
