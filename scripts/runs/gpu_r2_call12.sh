#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
pmc () { rm -rf $O/pmc_$1; timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_$1 -o p -- python scripts/attn_bench.py 3 0 > $O/pmc_$1.log 2>&1
  f=$(find $O/pmc_$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python scripts/pmc_summary.py "$f" attn_ >> $O/r2_attn_v2_pmc.txt || tail -5 $O/pmc_$1.log; }
: > $O/r2_attn_v2_pmc.txt
pmc a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
pmc b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
cat $O/r2_attn_v2_pmc.txt
timeout 200 python scripts/attn_bench.py 20 0 | tail -1
find $O -name "*.csv" -size +4M -delete
