#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "gemm or linear or adamw" 2>&1 | grep -E "passed|failed"
for rep in 1 2; do for v in prev new; do
  if [ $v == prev ]; then export MUSE_HIP_LIB=$PWD/open-muse_amd/muse/libmuse_hip_prevgemm.so; else unset MUSE_HIP_LIB; fi
  echo "--- $v"; WHICH=nn,nt,tt timeout 200 python scripts/gemm_probe.py 2>&1 | grep "linear"
done; done
for v in prev new; do
  if [ $v == prev ]; then export MUSE_HIP_LIB=$PWD/open-muse_amd/muse/libmuse_hip_prevgemm.so; else unset MUSE_HIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']; print('$v', d['value'], d['ms_per_step'], {k:v['ms_total'] for k,v in pk.items() if k.startswith('gemm_bf16')}, 'tr_ms', d['extra']['transformer_fwd_bwd_ms'])"
done
