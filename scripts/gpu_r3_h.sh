#!/bin/bash
# round 3, run H: how much of the plain step has no kernel in flight, and is the host the reason?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 200 python scripts/exp/host_bound.py full 2>&1 | grep -v amdgpu.ids | tee $O/r3h_host_bound.txt
timeout 200 python scripts/exp/host_bound.py tokens 2>&1 | grep -v amdgpu.ids | tee -a $O/r3h_host_bound.txt
rm -rf $O/prof_plain
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_plain -o t -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $O/r3h_plain.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/r3h_plain.txt | tail -1
f=$(find $O/prof_plain -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/overlap_report.py "$f" --last-ms 230 > $O/r3h_overlap_plain.txt 2>&1 && head -45 $O/r3h_overlap_plain.txt | cut -c1-150
[ -n "$f" ] && python - "$f" <<'PY' | tee $O/r3h_gaps.txt
# the idle gaps of the last ~4 steps: how long, and which kernels sit on either side of the longest ones
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - 230_000_000]
gaps, cur_end, last = [], rows[0][1], rows[0]
for r in rows[1:]:
    if r[0] > cur_end:
        gaps.append((r[0] - cur_end, last[2], r[2]))
    if r[1] > cur_end:
        cur_end, last = r[1], r
tot = sum(g[0] for g in gaps)
print(f"{len(gaps)} idle gaps, {tot / 1e6:.2f} ms of {(rows[-1][1] - rows[0][0]) / 1e6:.1f} ms; gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000)} = {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms; 5-20 us: {sum(g[0] for g in gaps if 5000 < g[0] <= 20000) / 1e6:.2f} ms; < 5 us: {sum(g[0] for g in gaps if g[0] <= 5000) / 1e6:.2f} ms")
from collections import Counter
c = Counter()
for g in gaps:
    c[(g[1][:40], g[2][:40])] += g[0]
for k, v in c.most_common(14):
    print(f"  {v / 1e6:7.3f} ms  after [{k[0]}] before [{k[1]}]")
PY
find $O/prof_plain -name "*kernel_trace*" -delete
