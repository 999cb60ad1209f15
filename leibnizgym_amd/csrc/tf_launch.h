// tf_launch.h - the seam between the C ABI host half (trifinger_hip.hip) and the fused step kernel (tf_env_kernels.hip).
//
// k_env<A, IS_RESET, ASYM, MODE, EXT, WIDE, HELP> has 2 x 2 x 8 instantiations per (EXT, WIDE) pair (2 x 2 x 4 with helper wavefronts: WIDE = 2);
// tf_env_kernels.hip is compiled once per pair (-DTF_EXT=0|1|2 -DTF_WIDE=0|1, -DTF_EXT=0|1|2 -DTF_WIDE=2) and exports one launcher each, so that the
// nine translation units build in parallel (make -j: ~1 min
// instead of ~4 for one translation unit).  Host side only: plain pointers and a stream.
#pragma once
#include <hip/hip_runtime.h>

#include "tf_params.h"

struct EnvLaunch {
    unsigned grid;               // workgroups = ceil(num_envs / 64)
    int action_dim;              // 9 or 18
    bool asym;                   // asymmetric observations: the states tile is emitted too
    const DevParams* d_params;   // device copy of the parameter block
    StepArgs sa;                 // what changes per launch, by value
    const float* action;         // [N][A] device tensor, or nullptr
    hipStream_t stream;
};

// which hooks of the reference step a launch performs (MODE of tf_roles.h)
enum { TF_LM_STEP = 0, TF_LM_STEP_RAND, TF_LM_RESET, TF_LM_RESETS, TF_LM_TORQUE, TF_LM_SIM, TF_LM_POST, TF_LM_FINISH };

// workgroups of the fused step (actions drawn in the launch) that fit a CU at once, as the HIP runtime computes it from registers and LDS
int tf_occupancy_env_0_0(int action_dim, bool asym);
int tf_occupancy_env_0_1(int action_dim, bool asym);
int tf_occupancy_env_1_0(int action_dim, bool asym);
int tf_occupancy_env_1_1(int action_dim, bool asym);
int tf_occupancy_env_2_0(int action_dim, bool asym);
int tf_occupancy_env_2_1(int action_dim, bool asym);
int tf_occupancy_env_0_2(int action_dim, bool asym);      // WIDE = 2: the 256-register kernels with helper wavefronts (TF_LM_STEP, _STEP_RAND, _RESET, _SIM)
int tf_occupancy_env_1_2(int action_dim, bool asym);
int tf_occupancy_env_2_2(int action_dim, bool asym);
void tf_launch_env_0_0(int lm, const EnvLaunch& a);      // tf_launch_env_<EXT>_<WIDE>
void tf_launch_env_0_1(int lm, const EnvLaunch& a);
void tf_launch_env_1_0(int lm, const EnvLaunch& a);
void tf_launch_env_1_1(int lm, const EnvLaunch& a);
void tf_launch_env_2_0(int lm, const EnvLaunch& a);
void tf_launch_env_2_1(int lm, const EnvLaunch& a);
void tf_launch_env_0_2(int lm, const EnvLaunch& a);
void tf_launch_env_1_2(int lm, const EnvLaunch& a);
void tf_launch_env_2_2(int lm, const EnvLaunch& a);
