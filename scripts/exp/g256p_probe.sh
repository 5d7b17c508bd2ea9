#!/bin/bash
# bring-up of csrc/gemm256p.h: correctness per epilogue mode, then timing A/B against the launch-per-tile kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
export MUSE_GEMM256=1
run() { env "$@" timeout 300 python scripts/exp/g256p_probe.py $MODE 2>&1 | grep -v amdgpu.ids | sed "s/^\[/[$* /"; }
MODE=check
{ run MUSE_G256P_EPI=1; run MUSE_G256P_EPI=2; run MUSE_G256P_EPI=2 MUSE_G256P_STAGGER=4 MUSE_G256P_STAUX=2; } > $O/g256p_check.txt 2>&1
grep -c "rerun_identical True" $O/g256p_check.txt; grep -E "False|e-0[01]|e\+0" $O/g256p_check.txt | cut -c1-200 | head -40
MODE=time
{ run MUSE_G256P=0; run MUSE_G256P_EPI=2; run MUSE_G256P_EPI=1; run MUSE_G256P_EPI=2 MUSE_G256P_STAGGER=2; run MUSE_G256P_EPI=2 MUSE_G256P_STAGGER=4;
  run MUSE_G256P_EPI=2 MUSE_G256P_STAUX=2; run MUSE_G256P_EPI=3; run MUSE_G256P=0; run MUSE_G256P_EPI=2; } > $O/g256p_time.txt 2>&1
cat $O/g256p_time.txt | cut -c1-200
