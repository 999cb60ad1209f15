#!/bin/bash
# Developer tool (GPU box), third part of the orientation study: the reference's dense rotation term (object_rot_delta) switched on, three seeds,
# 840 M frames each.   tools/orientation_study3.sh <tag> [epochs]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; E=${2:-3200}; O=gpurun_out; mkdir -p $O
for S in 7 11 23; do
  timeout 1500 python tools/ppo_learning_check.py $E 8192 $S fused 4 gym.reward_terms.object_rot_delta.activate=True 2>&1 | grep -v amdgpu.ids > $O/${T}_orientation_rot_delta_long_seed$S.txt
  grep -E "^epoch" $O/${T}_orientation_rot_delta_long_seed$S.txt | tail -1 | cut -c1-330; grep -E "play step  700" $O/${T}_orientation_rot_delta_long_seed$S.txt | cut -c1-300
done
