#!/bin/bash
# bench.py with edge values of the contract's flags (the driver chooses K and W)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for a in "--steps 1 --warmup 0" "--steps 2 --warmup 1" "--steps 3 --warmup 0 --no-prefetch"; do
  echo "== bench.py $a --no-extra --no-cpu-baseline"
  timeout 600 python bench.py $a --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
lines=[l for l in sys.stdin.read().strip().splitlines()]
print(len(lines), 'stdout line(s)')
d=json.loads(lines[-1]); print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','dtype')}, 'roofline' in d, d.get('cpu_baseline'))"
done
} > gpurun_out/bench_edge_args.txt 2>&1
cat gpurun_out/bench_edge_args.txt
