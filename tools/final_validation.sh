#!/bin/bash
# Developer tool (GPU box): what a round ends with after its last kernel change - GPU suite, smoke, soak (cube with DR, box), penetration statistics, the
# difficulty-4 learning checks (reference weights; dense rotation term), EXT profiles, the bench and its rocprofv3 summaries.   tools/final_validation.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu.txt 2>&1; tail -3 $O/${T}_pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python tools/soak.py 65536 100000 2>&1 | grep -v amdgpu.ids > $O/${T}_soak.txt; tail -2 $O/${T}_soak.txt
timeout 600 python tools/soak.py 65536 20000 box 2>&1 | grep -v amdgpu.ids > $O/${T}_soak_box.txt; tail -1 $O/${T}_soak_box.txt
{ timeout 600 python tools/penetration_stats.py 65536 3000; timeout 600 python tools/penetration_stats.py 65536 3000 dr; } 2>&1 | grep -v amdgpu.ids > $O/${T}_penetration.txt; cat $O/${T}_penetration.txt
timeout 500 python tools/ppo_learning_check.py 800 8192 7 fused 4 2>&1 | grep -v amdgpu.ids > $O/${T}_ppo_learning_d4_seed7.txt; grep "play step  700" $O/${T}_ppo_learning_d4_seed7.txt | head -1 | cut -c1-260
timeout 900 python tools/ppo_learning_check.py 3200 8192 23 fused 4 gym.reward_terms.object_rot_delta.activate=True 2>&1 | grep -v amdgpu.ids > $O/${T}_orientation_rot_delta_3200_seed23.txt; grep "play step  700" $O/${T}_orientation_rot_delta_3200_seed23.txt | head -1 | cut -c1-260
bash tools/profile_round.sh $T > /dev/null 2>&1
bash tools/profile_ext.sh $T > /dev/null 2>&1
python3 tools/sweep.py 2>&1 | grep -v amdgpu.ids > $O/${T}_sweep.txt; cat $O/${T}_sweep.txt
{ python3 tools/phase_timing.py 65536 600; python3 tools/phase_timing.py 8192 600; } 2>&1 | grep -v amdgpu.ids > $O/${T}_phase_timing_steady.txt
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('driver command: value %.4e ms_per_step %.4f kernel_avg_us %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us']))"; done | tee $O/${T}_driver_command.txt
head -c 400 $O/${T}_bench.json; echo; head -6 $O/${T}_kernel_trace_stats.txt
