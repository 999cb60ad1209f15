cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
python3 tools/phase_timing.py 65536 600 2>&1 | grep -v amdgpu.ids
python3 tools/phase_timing.py 65536 600 dr 2>&1 | grep -v amdgpu.ids
