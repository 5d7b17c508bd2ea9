#!/bin/bash
# fused GroupNorm statistics (conv_split epilogue, avgpool): kernel tests, VQGAN model tests, kernel-trace of a short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stats or split" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "vq or VQ or tokenizer" 2>&1 | tail -5
rm -rf $O/prof13
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof13 -o p -- python bench.py --steps 3 --warmup 2 --no-extra --no-cpu-baseline > $O/r2_call13_bench.log 2>&1
tail -1 $O/r2_call13_bench.log
f=$(find $O/prof13 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200 > $O/r2_call13_kernel_stats.txt; cat $O/r2_call13_kernel_stats.txt
find $O -name "*.csv" -size +4M -delete
