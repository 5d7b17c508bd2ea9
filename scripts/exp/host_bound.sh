cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for m in tokens full; do for w in 1 0; do MUSE_WGRAD_STREAM=$w timeout 200 python scripts/exp/host_bound.py $m 2>&1 | tail -1; done; done
