#!/bin/bash
# Developer tool (GPU box), second part of tools/orientation_study.sh: the reference's own DENSE rotation term (object_rot_delta, rewards.py:142-189;
# active in the env's default reward set, switched off by the difficulty-4 config) next to the 1 / (1 + angle) term.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; E=${2:-1600}; O=gpurun_out; mkdir -p $O
run() { name=$1; shift; timeout 1500 python tools/ppo_learning_check.py $E 8192 7 fused 4 "$@" 2>&1 | grep -v amdgpu.ids > $O/${T}_orientation_$name.txt; grep -E "^epoch" $O/${T}_orientation_$name.txt | tail -2 | cut -c1-330; grep -E "play step  700" $O/${T}_orientation_$name.txt | cut -c1-300; }
run rot_delta_on gym.reward_terms.object_rot_delta.activate=True
run rot_delta_x4 gym.reward_terms.object_rot_delta.activate=True gym.reward_terms.object_rot_delta.weight=-1000
run rot_delta_x4_no_rot gym.reward_terms.object_rot_delta.activate=True gym.reward_terms.object_rot_delta.weight=-1000 gym.reward_terms.object_rot.activate=False
