#!/bin/bash
# round 3, run K: same-box A/B lines of the round's tokenizer switches (one process each, default bench loop without the secondary legs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
for v in "MUSE_GN_FUSE=1 MUSE_CONV_IN_DIRECT=1" "MUSE_GN_FUSE=0 MUSE_CONV_IN_DIRECT=1" "MUSE_GN_FUSE=1 MUSE_CONV_IN_DIRECT=0" "MUSE_GN_FUSE=0 MUSE_CONV_IN_DIRECT=0" "MUSE_GN_FUSE=1 MUSE_CONV_IN_DIRECT=1"; do
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], 'images/s', d['ms_per_step'], 'ms')"
done | tee $O/r3k_tokenizer_ab.txt
