#!/bin/bash
# round 4: last check of the committed tree: full GPU suite, smoke, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/r4_last_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r4_last_pytest.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR" $O/r4_last_pytest.txt | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python bench.py > $O/r4_last_bench.json 2> $O/r4_last_bench.err; echo "bench exit $?"
python - <<PY
import json
d = json.loads([l for l in open("$O/r4_last_bench.json") if l.startswith("{")][-1])
e = d["extra"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "tr", e["transformer_mfma_frac"], e["transformer_fwd_bwd_ms"])
print("c5", e.get("vqgan_encode_decode_images_per_s_bf16x3"), "c4", e["config4_uvit_seq256"]["images_per_s"], e["config4_uvit_seq1024"]["images_per_s"], "x3", e["config4_uvit_seq256_bf16x3"]["images_per_s"], "f32", e["config4_uvit_seq256_f32"]["images_per_s"])
print("latency", {k: v for k, v in e["inference_latency"].items() if ("ms" in k or "error" in k) and "ref" not in k})
print(len(open("$O/r4_last_bench.json").read().strip().splitlines()), "stdout line(s)")
PY
