"""Fold a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE pass (scripts/gemm_probe.py, scripts/attn_bench.py)
into per-kernel clock and matrix-pipe-busy figures:  python scripts/pmc_fold.py <output dir of the pass>"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
cc = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
if not cc or not kt:
    print("no counter output")
    raise SystemExit
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("g256p", "g256::", "cslab", "cdma", "attn")):
        continue
    key = k[:70] + " grid=" + r["Grid_Size"]
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[key]["_us"].append(dur.get(r["Dispatch_Id"], float("nan")))
print("kernel | us (profiled) | GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration = clock GHz | SQ_VALU_MFMA_BUSY_CYCLES (summed over 1024 SIMDs) "
      "/ 1024 / cycles = matrix-pipe busy share")
for k, d in agg.items():
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    us, gui, mf = med(d["_us"]), med(d.get("GRBM_GUI_ACTIVE", [0])), med(d.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
    print(f"{k}\n    {us:8.1f} us  gui_active {gui:.4g} -> {gui / 8 / us / 1e3:.3f} GHz   mfma_busy {mf:.4g} -> {mf / 1024 / max(gui / 8, 1):.3f} of the "
          f"matrix-pipe cycles   wave_cycles {med(d.get('SQ_WAVE_CYCLES', [0])):.4g}  busy_cycles {med(d.get('SQ_BUSY_CYCLES', [0])):.4g}")
