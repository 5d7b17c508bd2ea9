#!/bin/bash
# round 2 checkpoint: smoke, bench line, rocprof kernel stats of the train step with the new attention
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r2_smoke.txt 2>&1; tail -2 $O/r2_smoke.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $O/r2_bench_a.txt 2>&1; tail -1 $O/r2_bench_a.txt | cut -c1-1500
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/r2_prof.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r02_run1_kernel_stats.csv && head -30 "$f" | cut -c1-170
find $O/prof -name "*kernel_trace*" -delete
