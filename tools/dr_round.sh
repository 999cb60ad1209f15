cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python -m pytest tests/test_parity_hip_vs_oracle.py tests/test_domain_randomization.py -x -q 2>&1 | tail -3
python3 tools/dr_cost.py 65536 2>&1 | grep -v amdgpu.ids; python3 tools/dr_cost.py 16384 2>&1 | grep -v amdgpu.ids
