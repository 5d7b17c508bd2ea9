#!/bin/bash
# round 3, validation B: the new model-level GPU tests (general / text-conditioned MaskGitTransformer, batch-64 parity, guided decode)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -rP -p no:cacheprovider -k "general or cc12m or benched_batch or text_guided or uvit or generate2" > $O/r3b_pytest.txt 2>&1; echo "pytest exit $?" >> $O/r3b_pytest.txt
grep -E "passed|failed|pytest exit|^FAILED|^ERROR|vs the reference|worst parameter" $O/r3b_pytest.txt | cut -c1-420 | tail -40
grep -B2 -A25 "^___" $O/r3b_pytest.txt | head -150
