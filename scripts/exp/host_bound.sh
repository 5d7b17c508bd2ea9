cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 200 python scripts/exp/host_bound.py full 2>&1 | tail -1
