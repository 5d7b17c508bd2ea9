#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) over scripts/gemm_probe.py for the LDS-DMA kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
run_pmc () { # name, counters, which
  rm -rf $O/pmc_$1
  REPS=2 WHICH=$3 timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_$1 -o p -- python scripts/gemm_probe.py > $O/pmc_$1.log 2>&1
  f=$(find $O/pmc_$1 -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY' | tee -a $O/pmc2_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    if not any(t in k for t in ("g256", "cdma", "conv_split", "gemm_kernel")): continue
    k += " grid=" + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in d.items(): print("    %-32s per dispatch %.5g  (n=%d)" % (c, v / max(1, cnt[(k, c)]), cnt[(k, c)]))
PY
  else tail -5 $O/pmc_$1.log; fi
}
: > $O/pmc2_summary.txt
run_pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" nn,nt,tt,conv
run_pmc mem "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" nn,nt,tt,conv
run_pmc misc "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" nn,nt,tt,conv
find $O -name "*.csv" -size +4M -delete
