#!/bin/bash
# Developer tool (GPU box): is the ORIENTATION half of the difficulty-4 goal learnable on this physics?  The reference's reward
# (scripts/rlg_hydra.py:140-182) at its own weights, then with the rotation term up-weighted / the success bonus on, and with more solver work
# (16 sweeps, temporal Gauss-Seidel) to separate the contact model from the training set-up.
#   tools/orientation_study.sh <tag> [epochs]   ->  gpurun_out/<tag>_orientation_*.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; E=${2:-1600}; O=gpurun_out; mkdir -p $O
run() { name=$1; shift; timeout 1500 python tools/ppo_learning_check.py $E 8192 7 fused 4 "$@" 2>&1 | grep -v amdgpu.ids > $O/${T}_orientation_$name.txt; grep -E "^epoch" $O/${T}_orientation_$name.txt | tail -2 | cut -c1-330; grep -E "play step  700" $O/${T}_orientation_$name.txt | cut -c1-300; }
run reference_weights
run rot_weight_x5 gym.reward_terms.object_rot.weight=10000
run rot_weight_x5_success_bonus gym.reward_terms.object_rot.weight=10000 gym.termination_conditions.success.activate=True
run rot_weight_x5_16_sweeps gym.reward_terms.object_rot.weight=10000 gym.sim.physx.num_position_iterations=16
