#!/bin/bash
# cached decode graph; full bench line with the taming leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sampling.py -x -q -m gpu -s -k "hip_graph" 2>&1 | tail -6
timeout 900 python bench.py > $O/r2_call16_bench.json 2> $O/r2_call16_bench.err; tail -c 6000 $O/r2_call16_bench.json
