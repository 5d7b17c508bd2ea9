#!/bin/bash
# U-ViT: bf16 operands written by AdaLN / norm backward, text-state cast cached: parity, timing, fresh kernel ranking
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -rf $O/prof27
timeout 900 python -m pytest tests/test_gpu_uvit.py -q -m gpu -p no:cacheprovider > $O/r2_call27_pytest.txt 2>&1; grep -E "passed|failed|rror" $O/r2_call27_pytest.txt | tail -5
timeout 300 python scripts/uvit_bench.py 64 3 bf16 256 adamw 2>&1 | tail -1
timeout 300 python scripts/uvit_bench.py 16 3 bf16 1024 adamw 2>&1 | tail -1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof27 -o uvit -- python scripts/uvit_bench.py 64 3 bf16 256 adamw > $O/r2_call27_uvit.txt 2>&1
f=$(find $O/prof27 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2_call27_uvit_kernel_stats.csv && head -30 "$f" | cut -c1-150
find $O/prof27 -name "*kernel_trace*" -size +8M -delete
