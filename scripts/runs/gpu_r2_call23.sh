#!/bin/bash
# wide ffn_mid kernels, U-ViT (side-stream dW, bf16 GLU chain, multi-tensor AdamW): parity, then A/B timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "ffn_mid" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_uvit.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -k "transformer or train_step" 2>&1 | tail -4
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
for w in 0 1 2 3; do
  MUSE_FFN_MID_WIDE=$w timeout 300 $B > $O/r2_call23_bench_w$w.json 2> $O/r2_call23_bench_w$w.err
  echo "ffn_mid_wide=$w: $(python - <<PY
import json
try:
    d=json.loads(open('$O/r2_call23_bench_w$w.json').read().strip().splitlines()[-1])
    hb=d['roofline']['hbm_bound_kernels']
    print(d['value'], d['ms_per_step'], 'ffn_bwd', hb['ffn_mid_bwd']['ms_total'], 'ffn_fwd', hb['ffn_mid_fwd']['ms_total'], 'tr_ms', d['extra'].get('transformer_fwd_bwd_ms'))
except Exception as e:
    print('FAILED', e); print(open('$O/r2_call23_bench_w$w.err').read()[-1500:])
PY
)"
done
for ws in 0 1; do echo "uvit wgrad_stream=$ws"; MUSE_WGRAD_STREAM=$ws timeout 300 python scripts/uvit_bench.py 64 3 bf16 256 adamw 2>&1 | tail -2; done
