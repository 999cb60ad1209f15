cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r4_g_pytest_gpu.txt 2>&1; tail -3 $O/r4_g_pytest_gpu.txt
{ python3 tools/phase_timing.py 65536 600; python3 tools/phase_timing.py 65536 600 dr; python3 tools/phase_timing.py 8192 600; } 2>&1 | grep -v amdgpu.ids > $O/r4_g_phase_timing_steady.txt
{ python3 tools/dr_cost.py 65536; python3 tools/dr_cost.py 16384; } 2>&1 | grep -v amdgpu.ids > $O/r4_g_dr_cost.txt; cat $O/r4_g_dr_cost.txt
bash tools/profile_ext.sh r4_g > /dev/null 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4_g_bench_driver_command.json 2>/dev/null; head -c 300 $O/r4_g_bench_driver_command.json; echo
