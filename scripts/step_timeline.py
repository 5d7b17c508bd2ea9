"""Millisecond-by-millisecond picture of ONE concurrent train step from a rocprofv3 kernel trace of bench.py (scripts/gpu.sh trace): per
HIP stream the busy share of each 1 ms bin and the kernel family that took most of it - where the step's critical path idles.

    python scripts/step_timeline.py <t_kernel_trace.csv> [--bin-ms 1.0] [--dump step.csv]"""
import argparse
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import overlap_report as R   # noqa: E402


def short(name):
    for k, v in (("conv_slab", "conv"), ("g256p::kernel<unsigned short, 0, 0>", "gemmNN"), ("g256p::kernel<unsigned short, 0, 1>", "gemmNT"),
                 ("g256p::kernel<float", "gemmNNf"), ("kernel_group", "dW"), ("attn2::fwd", "attF"), ("attn2::bwd", "attB"), ("ffn_mid_bwd", "ffnB"),
                 ("ffn_mid_fwd", "ffnF"), ("ln_bwd", "lnB"), ("ln_pair", "lnF2"), ("ln_fwd", "lnF"), ("adamw", "adam"), ("sum_multi", "sum"),
                 ("avgpool", "pool"), ("conv_in_direct", "convin"), ("gemm_kernel", "gemm128"), ("ce_", "ce"), ("emb", "emb"), ("gn_", "gn")):
        if k in name:
            return v
    return name.split("(")[0].split("::")[-1][:10]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--bin-ms", type=float, default=1.0)
    ap.add_argument("--dump", default=None)
    a = ap.parse_args()
    rows = R.load(a.trace)
    ms_rows = [r for r in rows if r[2].startswith("mask_sample_kernel")]
    common = collections.Counter(r[4] for r in ms_rows).most_common(1)[0][0]
    marks = [r[0] for r in ms_rows if r[4] == common]
    gaps = [b - x for x, b in zip(marks, marks[1:])]
    med = sorted(gaps)[len(gaps) // 2]
    idx = max(i for i, g in enumerate(gaps) if 0.8 * med < g < 1.2 * med)
    w0, w1 = marks[idx], marks[idx + 1]
    step = [r for r in rows if r[1] > w0 and r[0] < w1]
    streams = sorted({r[4] for r in step}, key=lambda s: -sum(min(r[1], w1) - max(r[0], w0) for r in step if r[4] == s))
    print(f"step of {(w1 - w0) / 1e6:.2f} ms, {len(step)} kernels, streams (busiest first): {streams}")
    if a.dump:
        with open(a.dump, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["start_us", "end_us", "stream", "kernel"])
            for r in step:
                w.writerow([f"{(r[0] - w0) / 1e3:.1f}", f"{(r[1] - w0) / 1e3:.1f}", r[4], short(r[2])])
    nb = int((w1 - w0) / 1e6 / a.bin_ms) + 1
    bw = a.bin_ms * 1e6
    for b in range(nb):
        t0, t1 = w0 + b * bw, w0 + (b + 1) * bw
        cells = []
        for s in streams:
            fam = collections.Counter()
            for r in step:
                if r[4] != s:
                    continue
                o = min(r[1], t1) - max(r[0], t0)
                if o > 0:
                    fam[short(r[2])] += o
            busy = sum(fam.values()) / bw
            top = ", ".join(f"{k} {v / bw:.2f}" for k, v in fam.most_common(3))
            cells.append(f"{busy:4.2f} [{top:38s}]")
        print(f"{b * a.bin_ms:5.1f} ms | " + " | ".join(cells))


if __name__ == "__main__":
    main()
