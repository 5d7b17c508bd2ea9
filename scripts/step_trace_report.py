"""Per-stream / per-kernel picture of the CONCURRENT train step from a rocprofv3 kernel trace of bench.py (scripts/gpu.sh trace):
the timed steps are found by their mask_sample_kernel launches (one per step), the last `--steps` whole steps are analysed.

    python scripts/step_trace_report.py <t_kernel_trace.csv> [--steps 5] [--serial profiles/r05_final_kernel_stats.csv]

With --serial (the rocprof --stats summary of the SERIAL bench) every kernel's concurrent duration is printed next to its serial one."""
import argparse
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import overlap_report as R   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--serial", default=None)
    a = ap.parse_args()
    rows = R.load(a.trace)
    ms_rows = [r for r in rows if r[2].startswith("mask_sample_kernel")]
    # the timed steps run on the step's high-priority stream, bench.py's serial profiling steps on the default stream: keep the
    # stream most mask_sample launches ran on, then the last `steps` whole steps between consecutive launches on it
    common = collections.Counter(r[4] for r in ms_rows).most_common(1)[0][0]
    marks = [r[0] for r in ms_rows if r[4] == common]
    if len(marks) < a.steps + 1:
        print("not enough steps in the trace")
        return
    gaps = [b - x for x, b in zip(marks, marks[1:])]
    med = sorted(gaps)[len(gaps) // 2]
    regular = [i for i, g in enumerate(gaps) if 0.8 * med < g < 1.2 * med]
    last = regular[-1]
    first = last
    while first - 1 in regular and last - first + 1 < a.steps:
        first -= 1
    w0, w1 = marks[first], marks[last + 1]
    n = last + 1 - first
    win = [r for r in rows if w0 <= r[0] < w1]
    print(f"{n} steps, {(w1 - w0) / 1e6 / n:.2f} ms per step, {len(win) / n:.0f} kernel launches per step")
    bys = collections.defaultdict(list)
    for s, e, name, q, st in win:
        bys[st].append((s, e))
    for st, iv in sorted(bys.items()):
        print(f"stream {st}: {len(iv) / n:6.0f} launches, busy {R.union_length(iv) / 1e6 / n:6.2f} ms per step")
    h = R.depth_histogram(win)
    print("kernels in flight (ms per step):", {k: round(v / 1e6 / n, 2) for k, v in sorted(h.items())})
    serial = {}
    if a.serial and os.path.exists(a.serial):
        for r in csv.DictReader(open(a.serial)):
            serial[r["Name"][:60]] = float(r["AverageNs"]) / 1e3
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, name, q, st in win:
        k = (st, name[:60])
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
    print(f"{'stream':6s} {'kernel':60s} {'launches':>8s} {'ms/step':>8s} {'avg us':>8s} {'serial us':>9s} {'stretch':>7s}")
    for (st, name), (c, us) in sorted(agg.items(), key=lambda x: -x[1][1])[:24]:
        avg = us / c
        ser = serial.get(name)
        print(f"{st:6s} {name:60s} {c / n:8.1f} {us / 1e3 / n:8.3f} {avg:8.1f} " + (f"{ser:9.1f} {avg / ser:7.2f}" if ser else f"{'':9s} {'':7s}"))


if __name__ == "__main__":
    main()
