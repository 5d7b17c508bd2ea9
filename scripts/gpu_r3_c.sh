#!/bin/bash
# round 3, run C: (1) the failed test of run B again, (2) stream-placement sweep of the default train step: stream
# creation order, priorities, CU-masked tokenizer / dW stream, under 4 and 8 hardware queues, (3) plain vs one-rank data-parallel
# launch line under both queue counts
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "general" 2>&1 | tail -3
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/exp/libstream_placement.so scripts/exp/stream_placement.hip || exit 1
rm -f $O/stream_placement.txt
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 400 python scripts/exp/stream_placement.py base order:psh order:hps prio:-1,0 prio:0,-1 2>&1 | grep -v Warning | tee -a $O/stream_placement.txt
done
timeout 400 python scripts/exp/stream_placement.py base mask:64x:p mask:32x:p mask:128x:p mask:64x:s 2>&1 | grep -v Warning | tee -a $O/stream_placement.txt
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/r3c_plain_q$q.json 2>/dev/null
  GPU_MAX_HW_QUEUES=$q timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/r3c_dp1_q$q.json 2>/dev/null
  python - <<PY
import json
for n in ("plain", "dp1"):
    try:
        d = json.loads([l for l in open("$O/r3c_%s_q$q.json" % n) if l.startswith("{")][-1])
        print("hw queues $q", n, d["value"], "img/s", d["ms_per_step"], "ms")
    except Exception as ex:
        print("hw queues $q", n, "no line:", ex)
PY
done
