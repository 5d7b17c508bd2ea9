#!/bin/bash
# patch-slab conv + wgrad side stream + 8-channel GroupNorm apply: parity tests, then A/B bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "conv2d or groupnorm or gn_" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider 2>&1 | tail -8
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
for cfg in "0 0 0" "1 0 0" "1 1 0" "1 1 1"; do
  set -- $cfg
  MUSE_CONV_SLAB=$1 MUSE_GN_SPLIT8=$2 MUSE_WGRAD_STREAM=$3 timeout 300 $B > $O/r2_call18_bench_$1$2$3.json 2> $O/r2_call18_bench_$1$2$3.err
  echo "slab=$1 gn8=$2 wgstream=$3: $(python - <<PY
import json
d=json.loads(open('$O/r2_call18_bench_$1$2$3.json').read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']; hb=d['roofline'].get('hbm_bound_kernels',{})
print(d['value'], d['ms_per_step'], 'conv', pk.get('conv_bf16x3_dma',{}).get('avg_us'), 'gn', hb.get('groupnorm_silu',{}).get('ms_total'), 'tr_ms', d['extra'].get('transformer_fwd_bwd_ms'))
PY
)"
done
