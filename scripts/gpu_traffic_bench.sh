#!/bin/bash
# HBM traffic (PMC) of every kernel of the train step: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass),
# kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes -> gpurun_out/r04_traffic.json (copy to profiles/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prefetch > $O/pmc_$c.log 2>&1
  echo "$c exit $?"
done
f=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && [ -n "$w" ] && python scripts/traffic_summary.py "$f" "$w" $O/r04_traffic.json || { tail -5 $O/pmc_FETCH_SIZE.log; tail -5 $O/pmc_WRITE_SIZE.log; }
find $O -name "*.csv" -size +4M -delete
