#!/bin/bash
# MFMA-busy / clock counters of the GroupNorm-fused convolution, launch-per-tile against persistent (round 6), on the probe's 64 x 128^2 x 128 -> 128 layer
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
: > $O/r06_conv_persist_pmc.txt
for p in 0 1; do
  rm -rf $O/pmc_cp
  MUSE_CONV_PERSIST=$p REPS=3 WHICH=conv timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_cp -o p -- python scripts/gemm_probe.py > $O/pmc_cp.log 2>&1
  echo "=== MUSE_CONV_PERSIST=$p" >> $O/r06_conv_persist_pmc.txt
  python scripts/pmc_fold.py $O/pmc_cp | grep -A1 "conv_slab" >> $O/r06_conv_persist_pmc.txt
done
rm -rf $O/pmc_cp
cat $O/r06_conv_persist_pmc.txt
