#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "golden or accumulation or adamw or train_step or full_batch" > $O/pytest_mid.txt 2>&1; tail -6 $O/pytest_mid.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_mid.txt 2>&1; tail -1 $O/bench_mid.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k, v) for k, v in d['roofline']['per_kernel'].items()]"
