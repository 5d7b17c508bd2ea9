#!/bin/bash
# HBM traffic (PMC) of the dominant kernels, separate passes as MI355X_MICROARCH.md prescribes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  REPS=3 WHICH=nn,conv timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python scripts/gemm_probe.py > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    k = r.get("Kernel_Name", "")[:60]
    if r["Counter_Name"] == sys.argv[2] and ("gemm_kernel" in k or "conv_split" in k): agg[k].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(sys.argv[2], k, "per-dispatch (last):", v[-1], "n=", len(v))
PY
done
find $O -name "*.csv" -size +4M -delete
