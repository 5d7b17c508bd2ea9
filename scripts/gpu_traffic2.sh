#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O; : > $O/traffic2.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python scripts/conv_traffic.py > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY' | tee -a $O/traffic2.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_dma" in r.get("Kernel_Name", "") and r["Counter_Name"] == sys.argv[2]]
vals = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], "conv_dma_kernel 16x256x256x128 (+residual, +GN stats): per launch", vals)
PY
done
find $O -name "*.csv" -size +4M -delete
