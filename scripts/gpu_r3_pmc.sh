#!/bin/bash
# round 3: PMC pass the round-2 verdict asked for - SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE (+ wave cycles) for the persistent 256^2
# GEMM and the patch-slab convolution, counters in their own run (kernel trace only), so that the clock under these kernels is a
# counter (GRBM_GUI_ACTIVE / duration) and not an inference from s_memtime
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
rm -rf $O/pmc_r3
REPS=3 WHICH=nn,nt,conv timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_r3 -o p -- python scripts/gemm_probe.py > $O/pmc_r3.log 2>&1
python - <<'PY' | tee $O/r3_pmc_mfma_clock.txt
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_r3/**/*counter_collection.csv", recursive=True)
kt = glob.glob("gpurun_out/pmc_r3/**/*kernel_trace.csv", recursive=True)
if not cc or not kt:
    print("no counter output"); raise SystemExit
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("g256p", "g256::", "cslab", "cdma")):
        continue
    key = k[:70] + " grid=" + r["Grid_Size"]
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[key]["_us"].append(dur.get(r["Dispatch_Id"], float("nan")))
print("kernel | us (profiled) | GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration = clock GHz | SQ_VALU_MFMA_BUSY_CYCLES (summed over 1024 SIMDs) / 1024 / cycles = matrix-pipe busy share")
for k, d in agg.items():
    med = lambda v: sorted(v)[len(v) // 2]
    us, gui, mf = med(d["_us"]), med(d.get("GRBM_GUI_ACTIVE", [0])), med(d.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
    print(f"{k}\n    {us:8.1f} us  gui_active {gui:.4g} -> {gui / 8 / us / 1e3:.3f} GHz   mfma_busy {mf:.4g} -> {mf / 1024 / max(gui / 8, 1):.3f} of the matrix-pipe cycles"
          f"   wave_cycles {med(d.get('SQ_WAVE_CYCLES', [0])):.4g}  busy_cycles {med(d.get('SQ_BUSY_CYCLES', [0])):.4g}")
PY
find $O/pmc_r3 -name "*.csv" -size +2M -delete
