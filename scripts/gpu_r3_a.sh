#!/bin/bash
# round 3, validation A: full GPU suite with the persistent GEMM as default, GEMM probe (strip skipping), same-box A/B of the default
# train step with and without the persistent kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r3a_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/r3a_pytest_gpu.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/r3a_pytest_gpu.txt | tail -8
run() { env "$@" timeout 300 python scripts/exp/g256p_probe.py time 2>&1 | grep -v amdgpu.ids | sed "s/^\[/[$* /"; }
{ run MUSE_G256P=0; run MUSE_G256P=1; } > $O/r3a_g256p_time.txt 2>&1
cat $O/r3a_g256p_time.txt | cut -c1-150
for p in 0 1 0 1; do
  MUSE_G256P=$p timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/r3a_bench_p$p.json 2> $O/r3a_bench_p$p.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/r3a_bench_p$p.json") if l.startswith("{")][-1])
    print("G256P=$p value", d["value"], "ms", d["ms_per_step"], "tr_frac", d.get("extra", {}).get("transformer_mfma_frac"))
except Exception as ex:
    print("no bench line:", ex)
PY
done
