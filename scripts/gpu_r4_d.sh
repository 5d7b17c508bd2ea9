#!/bin/bash
# round 4, run D: kernel-level profile of one U-ViT forward at the decoding batch (inference-latency leg), config-4 bf16x3 leg after the
# attention-core change, new tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "upsample2x or three_bf16 or bf16x3 or taming" > $O/r4d_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r4d_pytest.txt
grep -E "passed|failed|pytest exit" $O/r4d_pytest.txt | tail -3
rm -rf $O/prof_dec
BS=1 REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec -o dec -- python scripts/exp/decode_profile.py > $O/r4d_decode_prof.txt 2>&1
tail -1 $O/r4d_decode_prof.txt
f=$(find $O/prof_dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r4d_decode_kernel_stats.csv && head -30 "$f" | cut -c1-150
BS=1 REPS=20 python scripts/exp/decode_profile.py | tail -1
echo "uvit x3 $(python bench.py --uvit-leg 64,256,2,x3 2>/dev/null | tail -1 | cut -c1-120)"
