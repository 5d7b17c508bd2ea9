#!/bin/bash
# round 4, call t: the latency leg of bench.py with the 512 x 512 rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python bench.py --leg latency 2>/dev/null | tail -1 > gpurun_out/r4_t_latency.json
python -c "
import json; d=json.load(open('gpurun_out/r4_t_latency.json')); print({k:v for k,v in d.items() if 'ms' in k or 'error' in k}); print(d.get('model_512'))"
