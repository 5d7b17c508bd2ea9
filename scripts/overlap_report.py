"""Concurrency report of a rocprofv3 kernel trace: which HIP stream ran on which hardware queue, how busy each was, how much of the
wall time had 0 / 1 / 2+ kernels in flight, and how long each pair of kernel families really overlapped.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra
    python scripts/overlap_report.py out/**/t_kernel_trace.csv [--last-ms 300]

Made for DESIGN.md section 7 item 5 (the step time depends on the stream -> queue placement): the serial profiles under profiles/
say what each kernel costs alone, this says what the default, multi-stream step does with them."""
import argparse
import csv
import sys
from collections import defaultdict

FAMILIES = (("conv", ("conv_slab", "conv_dma", "conv_split", "conv")), ("gemm", ("g256::", "g256p::", "gemm_kernel")),
            ("attention", ("attn_",)), ("adamw", ("adamw",)), ("groupnorm/pool", ("gn_", "avgpool")),
            ("rows", ("ln_", "ffn_mid", "norm_res", "adaln", "glu_", "gelu", "silu", "grn_", "dwconv")),
            ("reduce", ("sum_slices", "colsum")), ("copy/cast", ("copyBuffer", "cast_", "fillBuffer", "elementwise")),
            ("collective", ("nccl", "rccl", "AllReduce")))


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def load(path, last_ms=None):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                         r.get("Stream_Id", "?")))
    rows.sort()
    if last_ms is not None and rows:
        t_end = max(r[1] for r in rows)
        rows = [r for r in rows if r[0] >= t_end - int(last_ms * 1e6)]
    return rows


def union_length(iv):
    """total length of the union of [start, end) intervals"""
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def depth_histogram(rows):
    """ns of wall time with k kernels in flight, k = 0, 1, 2, 3+ (between the first start and the last end)"""
    ev = []
    for s, e, *_ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist, depth, last = defaultdict(int), 0, ev[0][0]
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        depth += d
        last = t
    return hist


def merged(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def overlap_length(a, b):
    """length of the intersection of two unions of intervals (both given merged and sorted)"""
    i = j = tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def report(rows, out=sys.stdout):
    if not rows:
        print("no kernel dispatches", file=out)
        return {}
    t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
    span = t1 - t0
    print(f"{len(rows)} kernel dispatches over {span / 1e6:.2f} ms", file=out)
    # stream -> queue placement
    per = defaultdict(list)
    for s, e, name, q, st in rows:
        per[(st, q)].append((s, e))
    print("\nstream  queue  kernels  busy ms  share of span", file=out)
    for (st, q), iv in sorted(per.items(), key=lambda kv: -union_length(kv[1])):
        b = union_length(iv)
        print(f"{st:>6}  {q:>5}  {len(iv):7d}  {b / 1e6:7.2f}  {b / span:6.1%}", file=out)
    queues = defaultdict(set)
    for (st, q) in per:
        queues[q].add(st)
    shared = {q: sorted(s) for q, s in queues.items() if len(s) > 1}
    print("\nhardware queues shared by several streams:", shared if shared else "none", file=out)
    # kernels in flight
    hist = depth_histogram(rows)
    print("\nkernels in flight   ms     share", file=out)
    for k in range(4):
        print(f"   {'3+' if k == 3 else k:>2}            {hist[k] / 1e6:7.2f}  {hist[k] / span:6.1%}", file=out)
    # families
    fam = defaultdict(list)
    for s, e, name, q, st in rows:
        fam[family(name)].append((s, e))
    m = {f: merged(iv) for f, iv in fam.items()}
    names = sorted(m, key=lambda f: -sum(e - s for s, e in fam[f]))
    print("\nfamily            kernel-time ms   wall ms (union)   of which overlapped by another family", file=out)
    result = {"span_ms": span / 1e6, "idle_ms": hist[0] / 1e6, "families": {}, "shared_queues": shared}
    for f in names:
        ktime = sum(e - s for s, e in fam[f])
        wall = sum(e - s for s, e in m[f])
        others = merged([iv for g in names if g != f for iv in fam[g]])
        ov = overlap_length(m[f], others)
        result["families"][f] = {"kernel_ms": ktime / 1e6, "wall_ms": wall / 1e6, "overlapped_ms": ov / 1e6}
        print(f"{f:16s}  {ktime / 1e6:12.2f}   {wall / 1e6:12.2f}      {ov / 1e6:8.2f} ({ov / max(wall, 1):.0%})", file=out)
    print("\npairwise overlap (ms of wall time with both families running):", file=out)
    for i, f in enumerate(names):
        for g in names[i + 1:]:
            ov = overlap_length(m[f], m[g])
            if ov > 0.01 * span:
                print(f"   {f:14s} x {g:14s} {ov / 1e6:8.2f}", file=out)
    return result


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last-ms", type=float, default=None, help="only the last N ms of the trace (the timed steps)")
    a = ap.parse_args()
    report(load(a.trace, a.last_ms))
