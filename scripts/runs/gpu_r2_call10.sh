#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_models.py -q --tb=short -p no:cacheprovider -s -k "autocast or full_depth or bench_batch or vs_reference_golden" > $O/r2_parity_tests.txt 2>&1
grep -E "passed|failed|logits vs f32|VQ index|torch\.(float32|bfloat16) logits|Error|assert" $O/r2_parity_tests.txt | cut -c1-400 | tail -30
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/r2_bench_b.txt 2>&1; tail -1 $O/r2_bench_b.txt | cut -c1-6000
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/r2_bench_torchrun1.txt 2>&1; tail -1 $O/r2_bench_torchrun1.txt | cut -c1-300
bash scripts/gpu_traffic_bench.sh 2>&1 | tail -16
