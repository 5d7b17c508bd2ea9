#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -rf $O/prof_plain
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_plain -o t -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $O/r3h_plain.txt 2>&1
f=$(find $O/prof_plain -name "*kernel_trace.csv" | head -1)
python scripts/exp/gap_dump.py "$f" 230 10 | tee $O/r3h_gap_dump.txt | cut -c1-160
find $O/prof_plain -name "*kernel_trace*" -delete
