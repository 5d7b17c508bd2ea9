cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python scripts/exp/long_run.py 2>&1 | tail -1
