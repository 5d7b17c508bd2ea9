#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for st in 2 1; do echo "== MUSE_GEMM_STAGES=$st"; MUSE_GEMM_STAGES=$st python scripts/gemm_probe.py 2>&1 | grep -v amdgpu.ids; MUSE_GEMM_STAGES=$st python scripts/gemm_sweep.py 2>&1 | grep -E "muse_gemm|K=64:"; done
