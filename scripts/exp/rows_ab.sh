#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in f8_l8 f4_l8 f4_l4; do
  echo "== $v"; MUSE_HIP_LIB=$PWD/scripts/exp/lib/libmuse_$v.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
