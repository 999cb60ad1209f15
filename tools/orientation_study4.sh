#!/bin/bash
# Developer tool (GPU box), fourth part of the orientation study: three more seeds of the dense-rotation-term run, and seed 7 for twice as long.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; mkdir -p $O
run() { S=$1; E=$2; timeout 1500 python tools/ppo_learning_check.py $E 8192 $S fused 4 gym.reward_terms.object_rot_delta.activate=True 2>&1 | grep -v amdgpu.ids > $O/${T}_orientation_rot_delta_${E}_seed$S.txt
  grep -E "^epoch" $O/${T}_orientation_rot_delta_${E}_seed$S.txt | tail -1 | cut -c1-330; grep -E "play step  700" $O/${T}_orientation_rot_delta_${E}_seed$S.txt | cut -c1-300; }
run 1 3200; run 2 3200; run 3 3200; run 7 6400
