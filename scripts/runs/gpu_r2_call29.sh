#!/bin/bash
# same-box A/B: row kernels with the independent loads hoisted (ffn_mid_bwd a/b, ln_bwd dres) vs before
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "ffn_mid or layernorm or norm" 2>&1 | grep -E "passed|failed"
for rep in 1 2; do for v in prev new; do
  if [ $v == prev ]; then export MUSE_HIP_LIB=$PWD/open-muse_amd/muse/libmuse_hip_prevrow.so; else unset MUSE_HIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/r2_call29_bench_$v.json 2> $O/r2_call29_bench_$v.err
  python - <<PY
import json
d=json.loads(open('$O/r2_call29_bench_$v.json').read().strip().splitlines()[-1])
hb=d['roofline']['hbm_bound_kernels']
print('$v', d['value'], d['ms_per_step'], {k:(v['ms_total'],v['GBps']) for k,v in hb.items() if k in ('ffn_mid_bwd','layernorm_bwd','layernorm_fwd','ffn_mid_fwd')}, 'tr_ms', d['extra'].get('transformer_fwd_bwd_ms'))
PY
done; done
