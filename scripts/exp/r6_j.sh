#!/bin/bash
# round 6: persistent GEMM, bf16 epilogue through a wave-private LDS transposition (MUSE_G256P_EPI=7) against the register form (2): bits, then time
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
G=$O/r06_g256p_epi7.txt; : > $G
export MUSE_GEMM256=1
for e in 2 7; do
  MUSE_G256P_EPI=$e timeout 600 python scripts/exp/g256p_probe.py check 2>&1 | grep -v amdgpu.ids | grep "sha" | sed "s/^\[/[epi=$e /" > $O/r06_epi_check_$e.txt
done
if diff <(sed 's/epi=2//' $O/r06_epi_check_2.txt) <(sed 's/epi=7//' $O/r06_epi_check_7.txt) > /dev/null; then echo "epi 7 == epi 2: every output hash identical ($(wc -l < $O/r06_epi_check_7.txt) products), reruns identical: $(grep -c 'rerun_identical True' $O/r06_epi_check_7.txt)" >> $G; else echo "epi 7 DIFFERS from epi 2" >> $G; diff $O/r06_epi_check_2.txt $O/r06_epi_check_7.txt | head -10 >> $G; fi
run() { env "$@" timeout 300 python scripts/exp/g256p_probe.py time 2>&1 | grep -v amdgpu.ids | sed "s/^\[/[$* /" >> $G; }
run MUSE_G256P_EPI=2
run MUSE_G256P_EPI=7
run MUSE_G256P_EPI=2
run MUSE_G256P_EPI=7
run MUSE_G256P_EPI=3
grep -E "epi 7|total" $G
