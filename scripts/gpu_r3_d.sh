#!/bin/bash
# round 3, run D: (1) the remaining failing general-transformer test with its traceback, (2) where the per-bucket AdamW of the
# data-parallel path should run (MUSE_OPT_REDUCER_STREAM side | comm) under 4 and 8 hardware queues, one-rank launch line,
# (3) A/B of the wave-uniform MFMA-group skipping on the ragged row strip (scripts/exp/libmuse_skip.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q --tb=long -p no:cacheprovider -k "general and proj" 2>&1 | grep -v "^$" | tail -60 > $O/r3d_proj.txt; tail -45 $O/r3d_proj.txt | cut -c1-200
for where in side comm; do for q in 4 8; do
  MUSE_OPT_REDUCER_STREAM=$where GPU_MAX_HW_QUEUES=$q timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/r3d_dp1_${where}_q$q.json 2>/dev/null
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/r3d_dp1_${where}_q$q.json") if l.startswith("{")][-1])
    print("dp1 update on $where, hw queues $q:", d["value"], "img/s", d["ms_per_step"], "ms")
except Exception as ex:
    print("dp1 $where q$q no line:", ex)
PY
done; done
run() { env "$@" timeout 300 python scripts/exp/g256p_probe.py time 2>&1 | grep -v amdgpu.ids | sed "s/^\[/[$* /"; }
{ run A=1; run MUSE_HIP_LIB=scripts/exp/libmuse_skip.so; run A=2; run MUSE_HIP_LIB=scripts/exp/libmuse_skip.so; } > $O/r3d_skip_time.txt 2>&1
grep -E "FFN-in|QKV|logits|FFN-out|total" $O/r3d_skip_time.txt | cut -c1-150
for lib in "" scripts/exp/libmuse_skip.so "" scripts/exp/libmuse_skip.so; do
  MUSE_HIP_LIB=$lib timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lib=[$lib]', d['value'], 'img/s', d['ms_per_step'], 'ms', d['extra'].get('transformer_mfma_frac'))"
done
