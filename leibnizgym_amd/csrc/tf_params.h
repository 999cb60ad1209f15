// tf_params.h - device-side parameter block and launch arguments of the TriFinger kernels.
#pragma once
#include "tf_device_math.h"

#define WAVE 64
#define MAX_STATES 122

// ------------------------------------------------------------------------------------------------------
// device-side parameter block (kernel argument, lives in SGPRs / scalar cache)
// ------------------------------------------------------------------------------------------------------
// table rows
#define TAB_OFF 0
#define TAB_INV (MAX_STATES)
#define TAB_ACT_LO (2 * MAX_STATES)
#define TAB_ACT_HI (2 * MAX_STATES + 18)
#define TAB_KP (2 * MAX_STATES + 36)
#define TAB_KD (2 * MAX_STATES + 45)
#define TAB_KS (2 * MAX_STATES + 54)
#define TAB_FLOATS (2 * MAX_STATES + 63)

struct RewardCoef {
    float c_reach, c_move_pen, dt, c_dist, rot_num, rot_scale, w_rot, rot_delta_sched, w_rot_delta, w_move;
};

// Buffer pointers that are read out of the parameter block carry the global address space in their type: a plain
// pointer loaded from memory is "generic" to the compiler, which then emits flat_load/flat_store - those count
// against lgkmcnt as well as vmcnt, so every LDS read or scalar load that follows a store would wait for HBM.
#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS float gfloat;
typedef GLOBAL_AS uint8_t gu8;
typedef GLOBAL_AS int32_t gi32;
typedef GLOBAL_AS int64_t gi64;
typedef GLOBAL_AS uint32_t gu32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct DevParams {
    // buffers
    gfloat* state;
    gfloat* action_buf;
    gfloat* obs;
    gfloat* states;
    gfloat* reward;
    gu8* reset_buf;
    gu8* goal_reset_buf;
    gu8* successes;
    gu8* dones;
    gi64* steps;             // int64 like the reference's _steps_count_buf; the count itself fits 32 bits (episode lengths)
    gu32* reset_count;
    gfloat* info;
    gfloat* scratch;
    gu32* tickets;           // library-owned accumulators of the in-kernel statistics fold (STAT_* below)
    // sizes
    int32_t N, A, OD, SD;
    int32_t env_id_offset;
    uint32_t seed_lo, seed_hi;
    // MDP
    int32_t command_mode, normalize_action, normalize_obs, apply_safety_damping, asymmetric_obs, enable_ft;
    int32_t task_difficulty, episode_length;
    int32_t norm_p;                 // finger_reach_object_rate: p of the vector norm (2 in the shipped configs)
    int32_t robot_reset_type, object_reset_type, goal_rotation_activate;
    float dof_pos_stddev, dof_vel_stddev, goal_rate;
    int32_t dr_enable;
    float dr_cube_mass[2], dr_cube_size[2], dr_friction[2], dr_motor[2], dr_link_mass[2], dr_restitution[2];
    float dr_base_pos[3], dr_stage_pos[2], dr_friction_robot[2], dr_friction_object[2], dr_friction_stage[2];
    float dr_obs_noise;      // half-width of the observation noise; 0 when off (or when dr_enable is 0)
    float dr_action_repeat;  // probability of re-applying the previous step's torque; 0 when off
    float clip_obs, clip_act; // fused wrapper clipping (tf_set_clipping); FLT_MAX when off
    int32_t rew_active[6];
    int32_t success_activate;
    float success_bonus, pos_tol, ori_tol;
    // stepping
    int32_t substeps, iters, inner, control_decimation;      // iters = solver_iterations x solver_inner passes; the finger-only rows run on every inner-th
    float dt, hsub;
    float grav[3];
    TfModel m;
    float wall_s[3];         // slopes of the boundary profile between its knots: (wall_r[i+1] - wall_r[i]) / (wall_z[i+1] - wall_z[i])
    float wall_c[3], wall_sn[3];   // cos and sin of the slope angle of each segment: 1 / sqrt(1 + s^2), s / sqrt(1 + s^2) (fingertip - boundary contact)
    // obs/states offset and 1/range tables, action limits, PD gains (index = TAB_*).  Embedded so that every access
    // is a scalar load at a constant offset of the parameter block (a pointer member would be fetched per lane).
    float tables[TAB_FLOATS];
};

// what changes every launch travels by value; everything else is read through a pointer to constant
// device memory so that the ~200 scalars of DevParams are fetched (scalar cache) where they are used
// instead of being pinned in SGPRs for the whole kernel
struct StepArgs {
    RewardCoef rc;
    int32_t nsim;
    uint32_t frame;          // frame count after this launch (counter of the observation-noise draws); the LOW 32 bits of the handle's int64 count:
                             // frame-keyed draws repeat after 2^32 frames (include/trifinger.h)
    uint32_t frame0;         // frame count at the start of the control step (counter of the action-repeat draw)
};


enum { RNG_OBJECT = 0, RNG_GOAL_POS = 1, RNG_GOAL_QUAT = 2, RNG_GOAL_ANGVEL = 3, RNG_ROBOT = 4, RNG_DR = 9 /* and 10 */,
       RNG_OBS_NOISE = 16 /* .. 22, counter = frame count instead of reset count */, RNG_ACT_REPEAT = 24 /* counter = frame count */,
       RNG_ACTION = 32 /* .. 36: fused action source, counter = frame count */ };
DEV void rng4(const DevParams& P, uint32_t gid, uint32_t count, uint32_t tag, float u[4]) {
    rng4_key(P.seed_lo, P.seed_hi, gid, count, tag, u);
}
