#!/bin/bash
# Developer tool (GPU box): everything a round's closing profiles/ entry holds beyond tools/profile_round.sh, in one gpurun call.
#   tools/final_round.sh <tag>  ->  gpurun_out/<tag>_{sweep,api_layers,phase_timing}.txt, <tag>_bench_config{1,3}.json, PPO rate + trace, GEMM table
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out
bash tools/profile_round.sh $T > $O/${T}_profile_log.txt 2>&1
python3 tools/sweep.py 2>&1 | grep -v amdgpu.ids > $O/${T}_sweep.txt
python3 tools/api_bench.py 2>&1 | grep -v amdgpu.ids > $O/${T}_api_layers.txt
python3 bench.py --difficulty 1 --envs 8192 --no-cpu-baseline > $O/${T}_bench_config1.json 2>/dev/null
python3 bench.py --dr --envs 16384 --no-cpu-baseline > $O/${T}_bench_config3.json 2>/dev/null
[ -f leibnizgym_amd/csrc/libtrifinger_hip_timing.so ] && { python3 tools/phase_timing.py 65536; python3 tools/phase_timing.py 8192; } 2>&1 | grep -v amdgpu.ids > $O/${T}_phase_timing.txt
bash tools/ppo_profile.sh $T > /dev/null 2>&1
python3 tools/gemm_kernels_bench.py 8192 2>&1 | grep -v amdgpu.ids > $O/${T}_gemm_kernels.txt
tail -3 $O/${T}_ppo_rate.txt; cat $O/${T}_sweep.txt; head -c 600 $O/${T}_bench.json
