#!/bin/bash
# round-2 re-entry validation: full GPU suite, smoke, default bench line, rocprof of the no-extras bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r2_call17_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/r2_call17_pytest_gpu.txt
tail -5 $O/r2_call17_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r2_call17_smoke.txt 2>&1; echo "smoke exit $?" >> $O/r2_call17_smoke.txt; tail -3 $O/r2_call17_smoke.txt
timeout 900 python bench.py > $O/r2_call17_bench.json 2> $O/r2_call17_bench.err; echo "bench exit $?"; tail -c 9000 $O/r2_call17_bench.json
rm -rf $O/prof17
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof17 -o r2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/r2_call17_prof.txt 2>&1
f=$(find $O/prof17 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2_call17_kernel_stats.csv && head -40 "$f"
find $O/prof17 -name "*kernel_trace*" -size +8M -delete
