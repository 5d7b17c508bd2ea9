#!/bin/bash
# persistent patch-slab conv (MUSE_CONV_SLAB=2) + next-batch token prefetch: parity, then A/B bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
MUSE_CONV_SLAB=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv2d" 2>&1 | tail -5
MUSE_CONV_SLAB=2 timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -k "vq or train_step or taming" 2>&1 | tail -5
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
for cfg in "1 --no-prefetch" "2 --no-prefetch" "1 --prefetch" "2 --prefetch"; do
  set -- $cfg
  pf=$2; [ "$pf" == "--prefetch" ] && pf=""
  MUSE_CONV_SLAB=$1 timeout 300 $B $pf > $O/r2_call19_bench_$1$2.json 2> $O/r2_call19_bench_$1$2.err
  echo "slab=$1 $2: $(python - <<PY
import json
try:
    d=json.loads(open('$O/r2_call19_bench_$1$2.json').read().strip().splitlines()[-1])
    pk=d['roofline']['per_kernel']
    print(d['value'], d['ms_per_step'], 'conv', pk.get('conv_bf16x3_dma',{}).get('avg_us'), 'frac', d['roofline']['frac'], 'tr_ms', d['extra'].get('transformer_fwd_bwd_ms'))
except Exception as e:
    print('FAILED', e); print(open('$O/r2_call19_bench_$1$2.err').read()[-1500:])
PY
)"
done
