#!/bin/bash
# 256^2 GEMM with the DMA pieces spread over the MFMA groups (MUSE_G256_SPREAD=1): parity, micro-benchmark, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
MUSE_G256_SPREAD=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "gemm" 2>&1 | tail -3
for s in 0 1; do echo "--- spread $s"; MUSE_G256_SPREAD=$s WHICH=nn,nt,tt timeout 200 python scripts/gemm_probe.py 2>&1 | grep "linear"; done
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
for s in 0 1; do
  MUSE_G256_SPREAD=$s timeout 300 $B > $O/r2_call24_bench_s$s.json 2> $O/r2_call24_bench_s$s.err
  echo "spread=$s: $(python - <<PY
import json
try:
    d=json.loads(open('$O/r2_call24_bench_s$s.json').read().strip().splitlines()[-1])
    pk=d['roofline']['per_kernel']
    print(d['value'], d['ms_per_step'], {k:(v['ms_total'],v['tflops']) for k,v in pk.items() if k.startswith('gemm')}, 'tr_ms', d['extra'].get('transformer_fwd_bwd_ms'), d['extra'].get('transformer_mfma_frac'))
except Exception as e:
    print('FAILED', e); print(open('$O/r2_call24_bench_s$s.err').read()[-1500:])
PY
)"
done
