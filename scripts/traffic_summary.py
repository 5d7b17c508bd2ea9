"""Fold the two PMC passes of scripts/gpu.sh final (FETCH_SIZE, WRITE_SIZE counter_collection.csv) into
profiles/r02_traffic.json: per kernel, per launch, HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the counters are in KiB; on
gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read - /opt/skills/guides/MI355X_MICROARCH.md, section HBM).
    python scripts/traffic_summary.py fetch.csv write.csv out.json"""
import collections, csv, json, sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if k in write and fetch[k][1] and write[k][1]:
        f, w = fetch[k][0] / fetch[k][1], write[k][0] / write[k][1]
        out[k[:120]] = {"launches_sampled": fetch[k][1], "fetch_size_KiB_per_launch": round(f, 1), "write_size_KiB_per_launch": round(w, 1),
                        "hbm_bytes_per_launch": round((2 * f + w) * 1024)}
top = dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:40])
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prefetch`; "
                   "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, averaged over all launches of the kernel in the run (scripts/gpu.sh final)",
           "kernels": top}, open(sys.argv[3], "w"), indent=1)
for k, v in list(top.items())[:12]:
    print(f"{v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch x{v['launches_sampled']:4d}  {k[:90]}")
