#!/bin/bash
# same-box A/B: previous g256 epilogue (libmuse_hip_prevgemm.so) vs the rewrite; U-ViT + model tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_uvit.py tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -x > $O/r2_call26_pytest.txt 2>&1; grep -E "passed|failed|error" $O/r2_call26_pytest.txt | tail -3
for rep in 1 2; do
for v in prev new; do
  if [ $v == prev ]; then export MUSE_HIP_LIB=$PWD/open-muse_amd/muse/libmuse_hip_prevgemm.so; else unset MUSE_HIP_LIB; fi
  echo "--- $v"; WHICH=nn,nt,tt timeout 200 python scripts/gemm_probe.py 2>&1 | grep "linear"
done; done
for v in prev new; do
  if [ $v == prev ]; then export MUSE_HIP_LIB=$PWD/open-muse_amd/muse/libmuse_hip_prevgemm.so; else unset MUSE_HIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/r2_call26_bench_$v.json 2> $O/r2_call26_bench_$v.err
  python - <<PY
import json
d=json.loads(open('$O/r2_call26_bench_$v.json').read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']
print('$v', d['value'], d['ms_per_step'], {k:(v['ms_total'],v['tflops']) for k,v in pk.items() if k.startswith('gemm')}, 'tr_ms', d['extra'].get('transformer_fwd_bwd_ms'))
PY
  timeout 300 python scripts/uvit_bench.py 64 3 bf16 256 adamw 2>&1 | tail -1
done
