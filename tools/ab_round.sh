cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python -m pytest tests/test_parity_hip_vs_oracle.py tests/test_contact_scenarios.py tests/test_contact_lcp_reference.py -x -q -m gpu 2>&1 | tail -3
python3 tools/ab_bench.py leibnizgym_amd/csrc/variants/libtf_base.so leibnizgym_amd/csrc/libtrifinger_hip.so 2>&1 | grep -v amdgpu.ids
