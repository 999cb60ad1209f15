cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
