#!/bin/bash
# g256 epilogue rewrite (reads folded into the accumulators before the first store, LDS-only barriers): parity + timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "gemm or linear" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_uvit.py tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
WHICH=nn,nt,tt timeout 200 python scripts/gemm_probe.py 2>&1 | grep "linear"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/r2_call25_bench.json 2> $O/r2_call25_bench.err
python - <<PY
import json
d=json.loads(open('$O/r2_call25_bench.json').read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']
print(d['value'], d['ms_per_step'], {k:(v['ms_total'],v['tflops']) for k,v in pk.items() if k.startswith('gemm')}, 'tr_ms', d['extra'].get('transformer_fwd_bwd_ms'), d['extra'].get('transformer_mfma_frac'))
PY
timeout 300 python scripts/uvit_bench.py 64 3 bf16 256 adamw 2>&1 | tail -1
