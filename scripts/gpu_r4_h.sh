#!/bin/bash
# round 4, run H: bf16x3 GEMM mode as one launch over concatenated operands: tests, config-4 leg with one / three launches, latency leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_uvit.py -m gpu -q --tb=short -rP -p no:cacheprovider -k "three_bf16 or bf16x3 or config4" > $O/r4h_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r4h_pytest.txt
grep -E "passed|failed|pytest exit|^FAILED|Error" $O/r4h_pytest.txt | tail -5
grep -E "bf16x3 GEMM error|bf16x3 mode at|bf16x3 config 4" $O/r4h_pytest.txt | head -4
for v in 1 0 1; do echo "MUSE_X3_CAT=$v $(MUSE_X3_CAT=$v python bench.py --uvit-leg 64,256,2,x3 2>/dev/null | tail -1 | cut -c1-110)"; done | tee $O/r4h_x3_ab.txt
python bench.py --leg latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'ms' in k and 'ref' not in k})" | tee $O/r4h_latency.txt
