#!/bin/bash
# Developer tool (GPU box): everything a round's closing profiles/ entry holds, in one gpurun call.
#   tools/closing_round.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu.txt 2>&1; tail -3 $O/${T}_pytest_gpu.txt
bash tools/profile_round.sh $T > $O/${T}_profile_log.txt 2>&1
bash tools/profile_ext.sh $T > $O/${T}_ext_log.txt 2>&1
python3 tools/sweep.py 2>&1 | grep -v amdgpu.ids > $O/${T}_sweep.txt
python3 tools/api_bench.py 2>&1 | grep -v amdgpu.ids > $O/${T}_api_layers.txt
python3 bench.py --difficulty 1 --envs 8192 --no-cpu-baseline > $O/${T}_bench_config1.json 2>/dev/null
[ -f leibnizgym_amd/csrc/libtrifinger_hip_timing.so ] && { python3 tools/phase_timing.py 65536; python3 tools/phase_timing.py 8192; } 2>&1 | grep -v amdgpu.ids > $O/${T}_phase_timing.txt
{ timeout 600 python tools/penetration_stats.py 65536 3000; timeout 600 python tools/penetration_stats.py 65536 3000 dr; } 2>&1 | grep -v amdgpu.ids > $O/${T}_penetration.txt
for S in 7 11 23; do timeout 500 python tools/ppo_learning_check.py 800 8192 $S fused 4 2>&1 | grep -v amdgpu.ids > $O/${T}_ppo_learning_d4_seed$S.txt; done
timeout 300 python tools/ppo_learning_check.py 150 8192 7 fused 1 2>&1 | grep -v amdgpu.ids > $O/${T}_ppo_learning_d1_seed7.txt
timeout 900 python tools/soak.py 65536 100000 2>&1 | grep -v amdgpu.ids > $O/${T}_soak.txt
timeout 600 python tools/soak.py 65536 20000 box 2>&1 | grep -v amdgpu.ids > $O/${T}_soak_box.txt
bash tools/ppo_profile.sh $T > /dev/null 2>&1
tail -3 $O/${T}_ppo_rate.txt; cat $O/${T}_sweep.txt; head -c 400 $O/${T}_bench.json; echo; tail -3 $O/${T}_soak.txt; tail -5 $O/${T}_ppo_learning_d4_seed7.txt | cut -c1-250; cat $O/${T}_penetration.txt
