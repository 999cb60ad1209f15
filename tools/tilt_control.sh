#!/bin/bash
# Developer tool (GPU box): does the fingertip cone normal change what the learner reaches?  The dense-rotation-term run on NEW seeds with the library before
# that change (leibnizgym_amd/csrc/variants/libtf_pretilt.so, built from the parent commit) and with the shipped one.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; mkdir -p $O
for S in 4 5 6; do
  for V in pretilt shipped; do
    L=""; [ $V = pretilt ] && L=leibnizgym_amd/csrc/variants/libtf_pretilt.so
    TF_LIB=$L timeout 900 python tools/ppo_learning_check.py 3200 8192 $S fused 4 gym.reward_terms.object_rot_delta.activate=True 2>&1 | grep -v amdgpu.ids > $O/${T}_tilt_control_${V}_seed$S.txt
    echo "$V seed $S: $(grep 'play step  700' $O/${T}_tilt_control_${V}_seed$S.txt | head -1 | cut -c60-300)"
  done
done
