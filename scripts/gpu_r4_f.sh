#!/bin/bash
# round 4, run F: decoding graph kept across calls + small-batch split-K products: tests, latency leg in the four combinations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_uvit.py -m gpu -q --tb=short -p no:cacheprovider > $O/r4f_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r4f_pytest.txt
grep -E "passed|failed|pytest exit|^FAILED|Error" $O/r4f_pytest.txt | tail -6
cat > /tmp/lat.py <<'PY'
import json, os, sys
sys.argv = ["bench.py"]
sys.path.insert(0, os.getcwd())
import bench, torch
from muse import pipeline_muse
g = os.environ.get("GRAPH")
if g is not None:
    pipeline_muse.PipelineMuse.hip_graph = (g == "1")
print(json.dumps({k: v for k, v in bench.latency_leg(torch.device("cuda", 0)).items() if "ms" in k and "reference" not in k}))
PY
for sk in 1 0; do for gr in 1 0; do echo "SKINNY=$sk GRAPH=$gr $(MUSE_GEMM_SKINNY=$sk GRAPH=$gr python /tmp/lat.py 2>/dev/null | tail -1)"; done; done | tee $O/r4f_latency_ab.txt
