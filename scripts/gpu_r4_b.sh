#!/bin/bash
# round 4, run B: tests of the decoder kernels / bf16x3 GEMM mode / in-backward bucket reduction, same-box A/Bs (grouped dW split,
# decoder kernels, data-parallel launch line with and without the comparison leg), config-4 legs in f32 / bf16x3
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_uvit.py -m gpu -q --tb=short -rP -p no:cacheprovider \
  -k "conv_out_direct or upsample2x or three_bf16 or decoder_special or bf16x3 or vqgan or config4 or gradient_buckets or taming or grouped or fused_adamw" \
  > $O/r4b_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r4b_pytest.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/r4b_pytest.txt | tail -12
grep -E "bf16x3 GEMM error|bf16x3 mode at|decoder special|config 4 vs" $O/r4b_pytest.txt | head -8
for v in 3 1 2 5 7 3 1; do
  MUSE_WGRAD_GROUP=$v python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra 2>$O/r4b_bench_g$v.err | tail -1 > $O/r4b_bench_g$v.json
  python -c "import sys,json; d=json.loads(open('$O/r4b_bench_g$v.json').read()); e=d['extra']; print('WGRAD_GROUP=$v', d['value'], 'images/s', d['ms_per_step'], 'ms; transformer', e['transformer_fwd_bwd_ms'], 'ms frac', e['transformer_mfma_frac'], 'TT', d['roofline']['per_kernel'].get('gemm_bf16_TT'))" 2>&1 | tail -1
done | tee $O/r4b_wgrad_group_ab.txt
for v in "MUSE_CONV_OUT_DIRECT=1 MUSE_UPSAMPLE_SPLIT=1" "MUSE_CONV_OUT_DIRECT=0 MUSE_UPSAMPLE_SPLIT=0" "MUSE_CONV_OUT_DIRECT=1 MUSE_UPSAMPLE_SPLIT=0" "MUSE_CONV_OUT_DIRECT=1 MUSE_UPSAMPLE_SPLIT=1"; do
  echo "$v $(env $v python bench.py --leg vqgan,64 2>/dev/null | tail -1)"
done | tee $O/r4b_config5_ab.txt
for v in 0 1; do
  MUSE_BENCH_COMM_PLAIN=$v timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$v bench.py --gpus 1 --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>$O/r4b_dp1_$v.err | tail -1 > $O/r4b_dp1_$v.json
  python -c "import json; d=json.loads(open('$O/r4b_dp1_$v.json').read()); print('dp1 COMM_PLAIN=$v', d['value'], d['ms_per_step'], 'transformer', d['extra']['transformer_fwd_bwd_ms']); print(json.dumps(d.get('comm'))[:1800])"
done | tee $O/r4b_dp1.txt
ls /tmp/muse_rccl_* 2>/dev/null | head -3; for f in /tmp/muse_rccl_*; do head -12 "$f"; break; done
for spec in "64,256,2,x3" "32,256,2,f32" "128,256,3"; do
  echo "uvit $spec $(python bench.py --uvit-leg $spec 2>/dev/null | tail -1 | cut -c1-330)"
done | tee $O/r4b_config4.txt
