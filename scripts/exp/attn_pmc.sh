#!/bin/bash
# PMC passes over the fused attention kernels at the config-B shape (64 x 16 heads x 257 tokens x head_dim 48): which pipe is busy
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
run_pmc () { # name, counters
  rm -rf $O/apmc_$1
  timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/apmc_$1 -o p -- python ${ATTN_PMC_CMD:-scripts/attn_bench.py 5 0} > $O/apmc_$1.log 2>&1
  f=$(find $O/apmc_$1 -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY' | tee -a $O/attn_pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    if "attn" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in d.items(): print("    %-32s per dispatch %.5g  (n=%d)" % (c, v / max(1, cnt[(k, c)]), cnt[(k, c)]))
PY
  else tail -5 $O/apmc_$1.log; fi
}
: > $O/attn_pmc_summary.txt
run_pmc a "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
run_pmc b "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run_pmc c "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM"
find $O -name "*.csv" -size +4M -delete
