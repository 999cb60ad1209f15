#!/bin/bash
# Developer tool (GPU box): is the ORIENTATION half of the difficulty-4 goal learnable on this physics?  (results: profiles/r4_c_*, r4_e_*, r4_h_*)
#   tools/orientation_study.sh <tag> weights [epochs]   the reference's reward at its own weights, the 1/(1+angle) rotation term up-weighted five-fold,
#                                                      with the success bonus, with 16 sweeps
#   tools/orientation_study.sh <tag> dense [epochs]     the reference's own DENSE rotation term (object_rot_delta, rewards.py:142-189; active in the env's default
#                                                      reward set, switched off by the difficulty-4 config) on, x4, x4 without the 1/(1+angle) term
#   tools/orientation_study.sh <tag> solver [epochs]    the dense term on with 16 sweeps (seeds 11, 23) and with native.solver = tgs (seed 11)
#   tools/orientation_study.sh <tag> seeds [epochs]     the dense term on, seeds 7 11 23 1 2 3 (default 3200 epochs = 840 M frames each)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; PART=${2:-weights}; O=gpurun_out; mkdir -p $O
run() { name=$1; seed=$2; epochs=$3; shift 3; timeout 1500 python tools/ppo_learning_check.py $epochs 8192 $seed fused 4 "$@" 2>&1 | grep -v amdgpu.ids > $O/${T}_orientation_$name.txt
        grep -E "^epoch" $O/${T}_orientation_$name.txt | tail -1 | cut -c1-330; grep -E "play step  700" $O/${T}_orientation_$name.txt | cut -c1-300; }
case $PART in
  weights) E=${3:-1600}
    run reference_weights 7 $E
    run rot_weight_x5 7 $E gym.reward_terms.object_rot.weight=10000
    run rot_weight_x5_success_bonus 7 $E gym.reward_terms.object_rot.weight=10000 gym.termination_conditions.success.activate=True
    run rot_weight_x5_16_sweeps 7 $E gym.reward_terms.object_rot.weight=10000 gym.sim.physx.num_position_iterations=16 ;;
  dense) E=${3:-1600}
    run rot_delta_on 7 $E gym.reward_terms.object_rot_delta.activate=True
    run rot_delta_x4 7 $E gym.reward_terms.object_rot_delta.activate=True gym.reward_terms.object_rot_delta.weight=-1000
    run rot_delta_x4_no_rot 7 $E gym.reward_terms.object_rot_delta.activate=True gym.reward_terms.object_rot_delta.weight=-1000 gym.reward_terms.object_rot.activate=False ;;
  seeds) E=${3:-3200}
    for S in ${SEEDS:-7 11 23 1 2 3}; do run rot_delta_${E}_seed$S $S $E gym.reward_terms.object_rot_delta.activate=True; done ;;
  solver) E=${3:-3200}                       # does the solver residual matter for what is learned?  dense term on, 16 sweeps / temporal Gauss-Seidel against the shipped 8 sweeps
    run rot_delta_16_sweeps_seed11 11 $E gym.reward_terms.object_rot_delta.activate=True gym.sim.physx.num_position_iterations=16
    run rot_delta_16_sweeps_seed23 23 $E gym.reward_terms.object_rot_delta.activate=True gym.sim.physx.num_position_iterations=16
    run rot_delta_tgs_seed11 11 $E gym.reward_terms.object_rot_delta.activate=True gym.native.solver=tgs ;;
esac
