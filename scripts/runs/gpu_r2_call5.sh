#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
: > $O/r2_attn_nw.txt
for nw in 4 6 8; do echo "NW=$nw" >> $O/r2_attn_nw.txt; MUSE_ATT_NW=$nw timeout 300 python scripts/attn_bench.py 20 0,1 2>&1 | grep self >> $O/r2_attn_nw.txt; done
cat $O/r2_attn_nw.txt
