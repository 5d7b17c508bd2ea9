#!/bin/bash
# attention: tests + bench + SQ counters on the config-B shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q --tb=line -p no:cacheprovider -k "attention" > $O/r2_attn_tests.txt 2>&1
tail -3 $O/r2_attn_tests.txt
timeout 300 python scripts/attn_bench.py 20 > $O/r2_attn_bench.txt 2>&1; tail -6 $O/r2_attn_bench.txt
pmc () { rm -rf $O/pmc_$1; timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_$1 -o p -- python scripts/attn_bench.py 3 0 > $O/pmc_$1.log 2>&1
  f=$(find $O/pmc_$1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python scripts/pmc_summary.py "$f" attn_ >> $O/r2_attn_pmc.txt || tail -5 $O/pmc_$1.log; }
: > $O/r2_attn_pmc.txt
pmc a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
pmc b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"
pmc c "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
cat $O/r2_attn_pmc.txt
find $O -name "*.csv" -size +4M -delete
