"""Per-kernel, per-dispatch averages of a rocprofv3 counter_collection.csv:  python scripts/pmc_summary.py file.csv [kernel substring ...]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")
    if pats and not any(p in k for p in pats):
        continue
    k = k[:90] + " grid=" + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("    %-32s per dispatch %.5g  (n=%d)" % (c, v / max(1, cnt[(k, c)]), cnt[(k, c)]))
