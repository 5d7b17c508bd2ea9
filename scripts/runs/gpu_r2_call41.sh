#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python - <<'PY'
import sys, torch
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda:0")
for b, s in ((128, 256), (64, 1024), (128, 256), (64, 256)):
    r = bench.uvit_leg(dev, b, s, steps=3)
    print(b, s, r["ms_per_step"], r["mfma_frac"], r["peak_mem_GiB"], flush=True)
PY
