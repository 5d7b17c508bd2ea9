#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra"
for rep in 1 2; do for pr in none -1; do
  if [ $pr == none ]; then unset MUSE_MAIN_PRIORITY; else export MUSE_MAIN_PRIORITY=$pr; fi
  timeout 300 $B > $O/r2_call34_bench_$pr.json 2> $O/r2_call34_bench_$pr.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/r2_call34_bench_$pr.json').read().strip().splitlines()[-1])
    print('main prio $pr', d['value'], d['ms_per_step'])
except Exception as e:
    print('main prio $pr FAILED', open('$O/r2_call34_bench_$pr.err').read()[-800:])
PY
done; done
