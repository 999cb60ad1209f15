cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O; T=${1:-r3_c}
timeout 900 python -m pytest tests/test_parity_hip_vs_oracle.py tests/test_contact_scenarios.py -m gpu -x -q > $O/${T}_pytest_gpu.txt 2>&1; tail -3 $O/${T}_pytest_gpu.txt
python bench.py --no-cpu-baseline > $O/${T}_bench.json 2>$O/${T}_bench.err; head -c 300 $O/${T}_bench.json; echo
python bench.py --no-cpu-baseline --envs 8192 > $O/${T}_bench_8192.json 2>>$O/${T}_bench.err; head -c 300 $O/${T}_bench_8192.json; echo
{ python3 tools/phase_timing.py 65536; python3 tools/phase_timing.py 8192; } 2>&1 | grep -v amdgpu.ids > $O/${T}_phase_timing.txt; head -30 $O/${T}_phase_timing.txt
