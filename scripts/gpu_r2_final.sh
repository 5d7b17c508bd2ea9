#!/bin/bash
# round-2 validation of the committed tree: full GPU suite, smoke, PMC traffic passes, default bench line, rocprof of the serial bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r2_final_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/r2_final_pytest_gpu.txt
grep -E "passed|failed|pytest exit" $O/r2_final_pytest_gpu.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r2_final_smoke.txt 2>&1; echo "smoke exit $?" >> $O/r2_final_smoke.txt; tail -2 $O/r2_final_smoke.txt
bash scripts/gpu_traffic_bench.sh 2>&1 | tail -8
cp $O/r02_traffic.json profiles/r02_traffic.json 2>/dev/null
timeout 900 python bench.py > $O/r2_final_bench.json 2> $O/r2_final_bench.err; echo "bench exit $?"; tail -c 7000 $O/r2_final_bench.json
rm -rf $O/prof_final
MUSE_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-prefetch > $O/r2_final_prof.txt 2>&1
f=$(find $O/prof_final -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2_final_kernel_stats.csv && head -12 "$f" | cut -c1-160
find $O/prof_final -name "*kernel_trace*" -size +8M -delete
