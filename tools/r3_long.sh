cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 1500 python tools/ppo_learning_check.py 4000 8192 7 fused 4 2>&1 | grep -v amdgpu.ids > $O/r3_j_ppo_learning_d4_long_seed7.txt; grep -E "epoch (  49|1999|3999)|play step  700" $O/r3_j_ppo_learning_d4_long_seed7.txt | cut -c1-260
