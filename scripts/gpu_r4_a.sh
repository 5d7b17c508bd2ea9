#!/bin/bash
# round 4, run A: the new GPU tests (parameter-group AdamW, grouped weight gradients, VQ near-tie accounting, pipeline), same-box A/B of
# the grouped weight-gradient launch (MUSE_WGRAD_GROUP = 0 off / 1 / 2 / 3 slices), the driver's launch line at one rank (comm block),
# rocprof of the config-5 leg (encode -> decode)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/r4a_device.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_uvit.py tests/test_gpu_sampling.py -m gpu -q --tb=short -rP -p no:cacheprovider \
  -k "adamw or grouped or sum_multi or parameter_groups or groups_inside or vq_indices_over or streams_match or benched_batch or pipeline or general_vs or golden" \
  > $O/r4a_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r4a_pytest.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/r4a_pytest.txt | tail -12
grep -E "VQ index|image .* token" $O/r4a_pytest.txt | head -8
for v in 0 2 1 3 2 0; do
  MUSE_WGRAD_GROUP=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra 2>$O/r4a_bench_g$v.err | tail -1 > $O/r4a_bench_g$v.json
  python -c "import sys,json; d=json.loads(open('$O/r4a_bench_g$v.json').read()); e=d['extra']; print('WGRAD_GROUP=$v', d['value'], 'images/s', d['ms_per_step'], 'ms; transformer', e['transformer_fwd_bwd_ms'], 'ms frac', e['transformer_mfma_frac'], 'TT', d['roofline']['per_kernel'].get('gemm_bf16_TT'))" 2>&1 | tail -1
done | tee $O/r4a_wgrad_group_ab.txt
python -c "import json; d=json.loads(open('$O/r4a_bench_g2.json').read()); print(d['dtype']); print(d['extra']['measured_parity_bf16_vs_f32_mode'])"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 8 --warmup 3 --no-extra --no-cpu-baseline 2>$O/r4a_dp1.err | tail -1 > $O/r4a_dp1.json
python -c "import json; d=json.loads(open('$O/r4a_dp1.json').read()); print('dp1', d['value'], d['ms_per_step']); print(json.dumps(d.get('comm'))[:1500])"
rm -rf $O/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python bench.py --leg vqgan,64 > $O/r4a_c5_prof.txt 2>&1
f=$(find $O/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r4a_config5_kernel_stats.csv && head -16 "$f" | cut -c1-200
find $O/prof_c5 -name "*kernel_trace*" -size +8M -delete
tail -2 $O/r4a_c5_prof.txt
