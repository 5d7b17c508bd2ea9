#!/bin/bash
# round 6: what the persistent GEMM's tile change costs - epilogue forms (0 8-byte stores, 1 64-byte segments, 2 whole lines), no epilogue at all (3),
# and the whole-line epilogue with its store instructions removed (MUSE_G256P_STAUX=6): instruction stream vs memory
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
G=$O/r06_g256p_epilogue.txt; : > $G
export MUSE_GEMM256=1
run() { env "$@" timeout 300 python scripts/exp/g256p_probe.py time 2>&1 | grep -v amdgpu.ids | sed "s/^\[/[$* /" >> $G; }
run MUSE_G256P_EPI=2
run MUSE_G256P_EPI=2 MUSE_G256P_STAUX=6
run MUSE_G256P_EPI=3
run MUSE_G256P_EPI=1
run MUSE_G256P_EPI=1 MUSE_G256P_STAUX=6
run MUSE_G256P_EPI=0
run MUSE_G256P_EPI=2
grep -E "total" $G
