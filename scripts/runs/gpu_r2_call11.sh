#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=25 -x > $O/r2_pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/r2_pytest_gpu.txt
tail -45 $O/r2_pytest_gpu.txt | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/r2_bench_c.txt 2>&1; tail -1 $O/r2_bench_c.txt | cut -c1-3500
