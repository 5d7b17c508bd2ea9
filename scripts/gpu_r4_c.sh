#!/bin/bash
# round 4, run C: re-run of the tests the validation run failed (GroupNorm planes vs tensor after the SiLU rounding change), same-box A/B
# of the grouped weight gradients in the U-ViT leg, rocprof of the bf16x3 config-4 leg, the one-rank launch line with RCCL's INFO lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "split2_dma or groupnorm or conv or vqgan or bench_under" > $O/r4c_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r4c_pytest.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/r4c_pytest.txt | tail -6
for v in 1 0 3 1 0; do
  echo "uvit MUSE_WGRAD_GROUP=$v $(MUSE_WGRAD_GROUP=$v python bench.py --uvit-leg 128,256,3 2>/dev/null | tail -1 | cut -c1-140)"
done | tee $O/r4c_uvit_group_ab.txt
rm -rf $O/prof_x3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -o x3 -- python bench.py --uvit-leg 32,256,2,x3 > $O/r4c_x3_prof.txt 2>&1
f=$(find $O/prof_x3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r4c_config4_bf16x3_kernel_stats.csv && head -16 "$f" | cut -c1-190
find $O/prof_x3 -name "*kernel_trace*" -size +8M -delete
MUSE_BENCH_RCCL_DEBUG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>$O/r4c_dp1.err > $O/r4c_dp1.out
python -c "
import json
ls=[l for l in open('$O/r4c_dp1.out') if l.strip()]
print('stdout lines', len(ls)); d=json.loads(ls[-1]); print('dp1', d['value'], d['ms_per_step']); print(json.dumps(d.get('comm'))[:3000])"
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['ms_per_step'], d['extra']['transformer_fwd_bwd_ms'], d['extra']['transformer_mfma_frac'])"; done | tee $O/r4c_plain.txt
