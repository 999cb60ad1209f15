cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
for S in 11 3; do timeout 900 python tools/ppo_learning_check.py 3200 8192 $S fused 4 gym.reward_terms.object_rot_delta.activate=True 2>&1 | grep -v amdgpu.ids > $O/r4_n_orientation_rot_delta_3200_seed$S.txt; grep "play step  700" $O/r4_n_orientation_rot_delta_3200_seed$S.txt | head -1 | cut -c1-260; done
