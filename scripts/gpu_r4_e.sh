#!/bin/bash
# round 4, run E: the small-batch decoding path (split-K forward products): tests, forward time and inference-latency leg with it on / off
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sampling.py tests/test_gpu_uvit.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "small_batch or generate2 or pipeline or gemm or linear or uvit or golden" > $O/r4e_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r4e_pytest.txt
grep -E "passed|failed|pytest exit|^FAILED" $O/r4e_pytest.txt | tail -6
for v in 1 0 1 0; do echo "MUSE_GEMM_SKINNY=$v $(MUSE_GEMM_SKINNY=$v BS=1 REPS=30 python scripts/exp/decode_profile.py | tail -1) | $(MUSE_GEMM_SKINNY=$v BS=8 REPS=30 python scripts/exp/decode_profile.py | tail -1)"; done | tee $O/r4e_decode_ab.txt
for v in 1 0; do echo "MUSE_GEMM_SKINNY=$v $(MUSE_GEMM_SKINNY=$v python bench.py --leg latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'ms' in k})")"; done | tee $O/r4e_latency_ab.txt
rm -rf $O/prof_dec2
BS=1 REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec2 -o dec -- python scripts/exp/decode_profile.py > $O/r4e_decode_prof.txt 2>&1
f=$(find $O/prof_dec2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r4e_decode_kernel_stats.csv && head -8 "$f" | cut -c1-150
