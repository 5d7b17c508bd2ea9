#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "stale or pre_encoded_tokens or soft_code or general" 2>&1 | tail -30 | cut -c1-220
