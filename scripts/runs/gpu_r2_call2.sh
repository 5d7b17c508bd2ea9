#!/bin/bash
# round 2, call 2: the rewritten attention kernels: parity tests + micro-benchmark
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q --tb=short -p no:cacheprovider -k "attention" > $O/r2_attn_tests.txt 2>&1
echo "exit $?" >> $O/r2_attn_tests.txt; tail -40 $O/r2_attn_tests.txt
timeout 300 python scripts/attn_bench.py 20 > $O/r2_attn_bench.txt 2>&1; cat $O/r2_attn_bench.txt | tail -12
