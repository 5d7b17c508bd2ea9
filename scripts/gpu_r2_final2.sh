#!/bin/bash
# last validation of round 2 (7 GPU-minutes left): full GPU suite with the printed error figures, the default bench line with every
# leg (config 4 now on the cc12m geometry), smoke.  rocprof / PMC passes of this round stay those of scripts/gpu_r2_final.sh.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 330 python -m pytest tests -m gpu -q --tb=short -rP -p no:cacheprovider > $O/r2f2_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/r2f2_pytest_gpu.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/r2f2_pytest_gpu.txt | tail -12
grep -E "vs the reference|VQ index mismatches" $O/r2f2_pytest_gpu.txt | cut -c1-400
timeout 420 python bench.py > $O/r2f2_bench.json 2> $O/r2f2_bench.err; echo "bench exit $?"; tail -c 1500 $O/r2f2_bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r2f2_bench.json") if l.startswith("{")][-1])
    e = d["extra"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "tr_frac", e.get("transformer_mfma_frac"))
    print({k: v for k, v in e.items() if k.startswith("images_per_s") or k.startswith("config4")})
    print("cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "unit", "cores", "kind")})
except Exception as ex:
    print("no bench line:", ex)
PY
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/r2f2_smoke.txt 2>&1; echo "smoke exit $?" >> $O/r2f2_smoke.txt; tail -2 $O/r2f2_smoke.txt
# (if box time is left) kernel ranking of one config-4 step on the cc12m geometry, for the next round
rm -rf $O/prof_uvit
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_uvit -o uvit -- python bench.py --uvit-leg 64,256,2 > $O/r2f2_uvit_prof.txt 2>&1
f=$(find $O/prof_uvit -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2f2_uvit_kernel_stats.csv && head -14 "$f" | cut -c1-150
find $O/prof_uvit -name "*kernel_trace*" -delete
