#!/bin/bash
# round-3 validation of the committed tree: full GPU suite (with the printed error figures), smoke, PMC traffic passes, MFMA-busy /
# clock counters, the default bench line with every leg, rocprof of the serial bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r3_final}
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -rP -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_gpu.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/${T}_pytest_gpu.txt | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.txt 2>&1; echo "smoke exit $?" >> $O/${T}_smoke.txt; tail -2 $O/${T}_smoke.txt
bash scripts/gpu_traffic_bench.sh 2>&1 | tail -4
cp $O/r03_traffic.json profiles/r03_traffic.json 2>/dev/null
bash scripts/gpu_r3_pmc.sh 2>&1 | tail -14
timeout 1500 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench exit $?"; tail -c 1200 $O/${T}_bench.err
python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/${T}_bench.json") if l.startswith("{")][-1])
    e = d["extra"]
    print("value", d["value"], "ms", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "tr_frac", e.get("transformer_mfma_frac"))
    print({k: v for k, v in e.items() if k.startswith("images_per_s")})
    print("config4", e.get("config4_uvit_seq256"), e.get("config4_uvit_seq1024"))
    print("latency", e.get("inference_latency"))
    print("cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "unit", "cores", "kind", "cpu_model")})
except Exception as ex:
    print("no bench line:", ex)
PY
rm -rf $O/prof_final
MUSE_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-prefetch > $O/${T}_prof.txt 2>&1
f=$(find $O/prof_final -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats.csv && head -14 "$f" | cut -c1-170
find $O/prof_final -name "*kernel_trace*" -size +8M -delete
