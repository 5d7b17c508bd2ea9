#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
python scripts/gemm_probe.py > $O/probe_plain.txt 2>&1; cat $O/probe_plain.txt
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9a-z\[\]]+|TCP_[A-Z_0-9a-z]+|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|LdsBankConflict|OccupancyPercent|MemUnitStalled|L2CacheHit)\b" | sort -u > $O/counters.txt; wc -l $O/counters.txt
run_pmc () { # name, counters
  rm -rf $O/pmc_$1
  REPS=3 WHICH=$3 timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_$1 -o p -- python scripts/gemm_probe.py > $O/pmc_$1.log 2>&1
  f=$(find $O/pmc_$1 -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:70]
    if "gemm_kernel" not in k and "conv_split" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in d.items(): print("    %-32s %.4g (per dispatch %.4g)" % (c, v, v / max(1, cnt[(k, c)])))
PY
  else tail -5 $O/pmc_$1.log; fi
}
run_pmc sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" nn,conv
run_pmc sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" nn,conv
run_pmc tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" nn,conv
run_pmc grbm "GRBM_GUI_ACTIVE GRBM_COUNT" nn,conv
find $O -name "*.csv" -size +4M -delete
