#!/bin/bash
# L2 hit rate of the persistent GEMM's operand stream (round 6): rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum over scripts/gemm_probe.py (its own pass,
# kernel trace only), folded per kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
rm -rf $O/pmc_l2
REPS=3 WHICH=nn,nt,grp timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -o p -- python scripts/gemm_probe.py > $O/pmc_l2.log 2>&1
python - <<'PY' | tee gpurun_out/r06_gemm_l2_hit.txt
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_l2/**/*counter_collection.csv", recursive=True)
if not cc:
    print("no counter output"); print(open("gpurun_out/pmc_l2.log").read()[-1500:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("g256p", "g256::")):
        continue
    agg[k[:60] + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    med = lambda v: sorted(v)[len(v) // 2]
    h, m = med(d.get("TCC_HIT_sum", [0])), med(d.get("TCC_MISS_sum", [0]))
    print(f"{k}\n    TCC_HIT_sum {h:.4g}  TCC_MISS_sum {m:.4g}  -> L2 hit rate {h / max(h + m, 1):.3f}  ({(h + m) * 128 / 1e6:.0f} MB of 128-byte requests)")
PY
find $O/pmc_l2 -name "*.csv" -size +2M -delete
