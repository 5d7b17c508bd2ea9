#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
python -c "import torch; print('priority_range', torch.cuda.Stream.priority_range())"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "layernorm" 2>&1 | grep -E "passed|failed"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
for pr in none 1 0 -1; do
  if [ $pr == none ]; then unset MUSE_PREFETCH_PRIORITY; else export MUSE_PREFETCH_PRIORITY=$pr; fi
  timeout 300 $B > $O/r2_call33_bench_$pr.json 2> $O/r2_call33_bench_$pr.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/r2_call33_bench_$pr.json').read().strip().splitlines()[-1])
    hb=d['roofline']['hbm_bound_kernels']
    print('prio $pr', d['value'], d['ms_per_step'], 'ln_bwd', hb['layernorm_bwd']['ms_total'], 'ffn_bwd', hb['ffn_mid_bwd']['ms_total'])
except Exception as e:
    print('prio $pr FAILED', open('$O/r2_call33_bench_$pr.err').read()[-600:])
PY
done
