/*
 * trifinger.h - C ABI of the MI355X-native TriFinger vectorised environment step.
 *
 * This is the INNER drop-in boundary: the surface the reference's
 * task layer reaches through `isaacgym.gymapi` / `gymtorch`, re-cut as a plain C ABI so a
 * host in any language can bind it (ctypes stub: leibnizgym_amd/_capi.py).
 * No torch types appear here: every buffer is a raw device pointer owned by the caller.
 *
 * Two shared libraries export exactly these symbols with identical signatures:
 *   leibnizgym_amd/csrc/libtrifinger_hip.so   the product: hand-written HIP kernels for gfx950
 *   oracle/_build/libtrifinger_oracle.so      TEST INFRASTRUCTURE: scalar C restatement (pointers are
 *                                             host pointers, the stream argument is ignored)
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   tf_create / tf_destroy      gymapi.acquire_gym, create_sim, prepare_sim, destroy_sim
 *                               leibnizgym/envs/env_base.py:151,593,598,438
 *   tf_set_gravity              gymapi set_sim_params          env_base.py:175-193
 *   tf_set_clipping             VecTaskPython.step / get_state clamps (fused into the step)
 *                               leibnizgym/wrappers/vec_task.py:146-170
 *   tf_bind                     acquire_*_tensor + gymtorch.wrap_tensor (ownership inverted: the caller
 *                               allocates, the library receives pointers)
 *                               leibnizgym/envs/trifinger/trifinger_env.py:594-617
 *   tf_step                     IsaacEnvBase.step body: masked _reset_impl/_goal_reset_impl, _pre_step,
 *                               control_decimation x simulate, _post_step, step counters, timeout, dones
 *                               env_base.py:370-399; trifinger_env.py:373-559,959-1265
 *   tf_reset                    IsaacEnvBase.reset body        env_base.py:322-343
 *   tf_frame_count              gymapi.get_frame_count         env_base.py:289
 *   tf_apply_resets             set_dof_state_tensor_indexed / set_actor_root_state_tensor_indexed (mask
 *                               based, no index lists, no host sync)   trifinger_env.py:419-423,439
 *   tf_pre_step                 TrifingerEnv._pre_step + set_dof_actuation_force_tensor  trifinger_env.py:442-498
 *   tf_simulate                 gymapi.simulate (+fetch_results)       env_base.py:336-339,383-387
 *   tf_post_step                TrifingerEnv._post_step + refresh_*_tensor   trifinger_env.py:500-559,959-994
 *   tf_finish_step              env_base.py:391-399 (steps += 1, timeout, dones)
 *
 * Conventions: every entry returns a TfStatus (0 ok, negative = error; the Python host maps them to the
 * exception types the reference raises).  No entry allocates device memory or synchronises after tf_bind.
 * One handle <-> one stream <-> one GPU; handles are thread-compatible, not thread-safe.
 */
#ifndef TRIFINGER_H_
#define TRIFINGER_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TF_API_VERSION 8

typedef enum TfStatus {
    TF_OK = 0,
    TF_ERR_INVALID_ARG = -1,     /* NULL pointer, non-positive size ...                          */
    TF_ERR_COMMAND_MODE = -2,    /* ValueError  trifinger_env.py:476-478,649-651                 */
    TF_ERR_ROBOT_RESET = -3,     /* ValueError  trifinger_env.py:1142-1144                       */
    TF_ERR_OBJECT_RESET = -4,    /* ValueError  trifinger_env.py:1174-1177                       */
    TF_ERR_DIFFICULTY = -5,      /* ValueError  trifinger_env.py:1244-1246                       */
    TF_ERR_NOT_BOUND = -6,       /* tf_step before tf_bind                                       */
    TF_ERR_DEVICE = -7,          /* HIP runtime error (tf_last_error_string has the text)        */
    TF_ERR_UNSUPPORTED = -8      /* valid in the reference, not built yet                        */
} TfStatus;

/* command_mode (trifinger_env.py:36-37,460-478) */
enum { TF_CMD_TORQUE = 0, TF_CMD_POSITION = 1, TF_CMD_POSITION_IMPEDANCE = 2 };
/* reset_distribution.*.type (trifinger_env.py:48-68) */
enum { TF_RESET_NONE = 0, TF_RESET_DEFAULT = 1, TF_RESET_RANDOM = 2 };

/* reward terms, in the evaluation order of trifinger_env.py:513-550 */
enum {
    TF_REW_FINGER_REACH_OBJECT_RATE = 0,
    TF_REW_FINGER_MOVE_PENALTY = 1,
    TF_REW_OBJECT_DIST = 2,
    TF_REW_OBJECT_ROT = 3,
    TF_REW_OBJECT_ROT_DELTA = 4,
    TF_REW_OBJECT_MOVE = 5,
    TF_NUM_REWARD_TERMS = 6
};

/* Rows of the structure-of-arrays state matrix: float state[TF_STATE_ROWS][num_envs]
 * (row-major; env i of row r is state[r * num_envs + i]).  Quaternions are xyzw
 * (leibnizgym/utils/torch_utils.py:99-111). */
enum {
    TF_S_Q = 0,          /*  9 joint positions   (finger 0: 0..2, finger 120: 3..5, finger 240: 6..8) */
    TF_S_QD = 9,         /*  9 joint velocities                                                   */
    TF_S_CUBE_P = 18,    /*  3 cube position                                                      */
    TF_S_CUBE_Q = 21,    /*  4 cube orientation                                                   */
    TF_S_CUBE_V = 25,    /*  3 cube linear velocity                                               */
    TF_S_CUBE_W = 28,    /*  3 cube angular velocity                                              */
    TF_S_GOAL_P = 31,    /*  3 goal position      (_object_goal_poses_buf, trifinger_env.py:588)  */
    TF_S_GOAL_Q = 34,    /*  4 goal orientation                                                   */
    TF_S_GOAL_W = 38,    /*  3 goal angular velocity (_object_goal_movement_buf[:,3:6], :590)     */
    TF_S_TIP_P = 41,     /*  9 fingertip positions of the last filled frame (history[0][:, :, 0:3]) */
    TF_S_TAU = 50,       /*  9 applied joint torque (what set_dof_actuation_force_tensor received) */
    TF_S_PREV_OBJ_P = 59,/*  3 object position of history[1]: the pose the step's physics started from */
    TF_S_PREV_OBJ_Q = 62,/*  4 object orientation of history[1]                                   */
    TF_S_FT = 66,        /* 18 fingertip contact wrench (world frame force 3 + torque 3 per finger about
                              the tip-link origin, mean over the substeps of the step); split path only */
    TF_S_DR = 84,        /* 14 per-env domain-randomisation values drawn at reset (index list: TF_DR_* below): scale factors
                              (1.0 when DR is off) for cube mass, cube size, contact friction, motor torque, finger link
                              mass, finger contact restitution; offsets in metres (0.0 when off) of the robot base (x, y, z)
                              and of the stage = table + boundary (x, y); friction factors of the robot, the object and
                              the stage                                                                      */
    /* Warm start of the contact solver (an implementation choice, not algorithmic traffic: SURVEY 8d).  The impulses
     * of the last substep seed the next one when the contact kept its identity; a reset clears them.               */
    TF_S_LAM_FC = 98,    /* 12 finger-cube contact of finger f at [4f..4f+3]: normal impulse, world friction impulse (3) */
    TF_S_FC_LINK = 110,  /*  3 activity code of finger f's warm-start rows: link that held the finger-cube contact (0 none, 1 upper, 2 middle,
                              3 distal) + 4 if the fingertip-boundary contact pushed; the rows of an inactive contact (TF_S_LAM_FC of a
                              finger with link 0, TF_S_LAM_TW without the 4) are neither read nor written: content undefined          */
    TF_S_LAM_TF = 113,   /*  9 fingertip-floor contact of finger f at [3f..3f+2]: normal, tangent 1, tangent 2       */
    TF_S_LAM_TW = 122,   /*  9 fingertip-boundary-wall contact of finger f, same layout                             */
    TF_S_LAM_CF = 131,   /* 12 cube corner i against the floor at [3i..3i+2]: +z (normal), +x, +y                   */
    TF_S_CF_FACE = 143,  /*  1 cube face whose corners those were (0 none, 1..6)                                   */
    TF_S_LAM_CW = 144,   /* 12 cube corner i against the boundary wall at [3i..3i+2]: normal, tangent, +z          */
    TF_S_CW_FACE = 156,  /*  1 feature those rows belong to: cube face 1..6 + 8 x which pair of its corners is the lower one (0..3) + 32 x order
                          *    inside that pair (nearer corner first); 0 while no corner touches the boundary (TF_S_LAM_CW then undefined) */
    /* Samples of the env's NEXT reset (API 6; an implementation choice like the warm start, not algorithmic traffic).  The step that flags an env for
     * reset (time-out, non-finite state) draws the object pose and the goal of that reset at its END - on the finger wavefronts, which have finished by
     * then - and the reset, which sits at the start of every launch's critical path, only loads them.  The draws are functions of (seed, global env id,
     * reset count), so the values are the ones the reset would have drawn itself: TF_S_NEXT_TAG holds reset count + 1 as an integer bit pattern
     * (0: nothing stored), and a reset whose count does not match (goal resets in between, a reset flagged from outside) draws as before. */
    TF_S_NEXT_OBJ = 157, /*  4 object of the next reset: x, y relative to the stage centre, yaw quaternion z, w      */
    TF_S_NEXT_GOAL = 161,/* 10 goal of the next reset: position relative to the stage centre (3), orientation (4), angular velocity (3) */
    TF_S_NEXT_TAG = 171, /*  1 reset count + 1 the two were drawn for, as a uint32 bit pattern                       */
    TF_STATE_ROWS = 172
};
#define TF_NUM_DR 14
enum { TF_DR_CUBE_MASS = 0, TF_DR_CUBE_SIZE = 1, TF_DR_FRICTION = 2, TF_DR_MOTOR = 3, TF_DR_LINK_MASS = 4, TF_DR_RESTITUTION = 5,
       TF_DR_BASE_POS = 6 /* 6..8 */, TF_DR_STAGE_POS = 9 /* 9..10 */, TF_DR_FRICTION_ROBOT = 11, TF_DR_FRICTION_OBJECT = 12,
       TF_DR_FRICTION_STAGE = 13 };
/* value of DR slot j when the feature is off: offsets 0, factors 1 */
#define TF_DR_NEUTRAL(j) (((j) >= TF_DR_BASE_POS && (j) < TF_DR_FRICTION_ROBOT) ? 0.0f : 1.0f)
#define TF_NORM_INF (-1)      /* finger_reach_norm_p: the maximum norm */
/* Largest num_envs of one handle: the kernels address state[TF_STATE_ROWS][num_envs] with 32-bit byte offsets
 * (172 * 2 Mi * 4 B = 1.4 GB).  Larger populations are sharded over several handles / GPUs (env_id_offset). */
#define TF_MAX_ENVS 2097152

#define TF_OBS_DIM_BASE 32    /* 9 + 9 + 7 + 7; the action slot (9 or 18) follows  trifinger_env.py:280-286 */
#define TF_STATES_EXTRA 72    /* 6 + 39 + 9 + 18                                   trifinger_env.py:296-300 */
#define TF_NUM_INFO 16

/* info[] slots written by every step (device floats, no host sync):
 *   0..5  mean of reward term k over envs (env/rewards/<term>, only meaningful when active)
 *   6     env/current_position_goal/count        7  env/current_orientation_goal/count
 *   8     env/average_consecutive_success        9  number of envs reset in this step (diagnostic)
 *   10    number of envs with a non-finite state caught by the NaN guard (diagnostic)
 */
enum { TF_INFO_REW0 = 0, TF_INFO_POS_COUNT = 6, TF_INFO_ORI_COUNT = 7, TF_INFO_SUCCESS_MEAN = 8,
       TF_INFO_NUM_RESETS = 9, TF_INFO_NUM_NONFINITE = 10 };

/* Collision shape of a finger link, in the frame of the link: a TAPERED ROUNDED BOX swept along the axis a -> b.  At parameter s in
 * [0, 1] of the axis the cross-section is a rectangle with half widths w1(s), w2(s) along two fixed directions of the link frame
 * (upper link: x and z; middle and lower link: x and y), corner rounding rho(s) <= w and centre offset (o1(s), o2(s)) from the axis;
 * every quantity is linear in s and given at s = 0 and s = 1.  w1 = w2 = rho is a capsule.  Distance to the cube: the exact closest
 * points of the AXIS and the cube give s, the distance D and the unit direction u from the axis point to the cube point; the gap is
 * D - [(w1 - rho) |u1| + (w2 - rho) |u2| + rho + o1 u1 + o2 u2] - the support function of the cross-section along u: exact against
 * a face of the cube, a few millimetres early against an edge or a corner in a diagonal direction of the cross-section.          */
typedef struct TfLinkShape { float a[3], b[3], w1[2], w2[2], rho[2], o1[2], o2[2]; } TfLinkShape;
typedef struct TfSphere { float c[3], radius; } TfSphere;

typedef struct TfRewardTerm {
    int32_t activate;
    float weight;
    double sched_start;      /* thresh_sched_* (or linear_schedule_* for object_rot_delta); rewards.py:45-47 */
    double sched_end;
} TfRewardTerm;

/* Physical model: the build-authored spec of what the reference asks IsaacGym to simulate
 * (SURVEY.md section 8a-P).  tf_default_model() fills it from the URDF numbers. */
typedef struct TfModel {
    /* finger kinematic chain, identical for the three fingers (trifingerpro.urdf:161-190,461-475) */
    float base_height;            /* 0.29   base_to_upper_holder_joint                              */
    float base_yaw_cos[3];        /* yaw {0, -120deg, -240deg}                                      */
    float base_yaw_sin[3];
    float base_half_yaw_cos[3];   /* cos/sin of yaw/2 (fingertip orientation quaternion)            */
    float base_half_yaw_sin[3];
    float j2_origin[3];           /* (0.01685, 0.0505, 0)      in upper frame                        */
    float j3_origin[3];           /* (0.04922, 0, -0.16)       in middle frame                       */
    float tip_origin[3];          /* (0.0185, 0, -0.1626)      in lower frame (finger_tip_link)      */
    /* links 1..3 = upper, middle, lower(+tip merged).  inertia = xx,yy,zz,xy,xz,yz about the COM    */
    float link_mass[3];
    float link_com[3][3];
    float link_inertia[3][6];
    /* joint properties (trifinger_env.py:149-158,774-787) */
    float q_lo[3], q_hi[3];
    float qd_max;                 /* 10 rad/s                                                        */
    float tau_max;                /* 0.36 Nm                                                         */
    float link_angular_damping;   /* 0.01 (trifinger_env.py:866)                                     */
    float q_default[3];           /* (0, 0.9, -1.7)                                                  */
    /* collision primitives (build's choice): every finger link is a tapered rounded box (+ a sphere per joint housing) that covers the
     * convex hull the reference loads for it (meshes/stl/pro/SIM__BL-Finger_{Proximal,Intermediate,Tip_without_tip,Tip_actual_tip}.obj
     * with the collision origins of trifingerpro.urdf:88-153; one hull per link, trifinger_env.py:859-879) to within 3 mm, fitted by
     * tools/fit_link_shapes.py to the hull vertices of tests/golden/model.npz; tests/test_model_fixture.py holds the coverage.     */
    float cap_a[3], cap_b[3];     /* FINGERTIP capsule of the distal link (lower frame): the tube, b = centre of the fingertip sphere:
                                   * the shape of the fingertip-floor / fingertip-wall / finger-finger contacts and the axis of shape3 */
    float cap_radius;             /* 0.0102: radius of the fingertip sphere (SIM__BL-Finger_Tip_actual_tip.obj)                   */
    /* the shapes of the three links (include/trifinger_default_caps.h, fitted to the hulls by tools/fit_link_shapes.py); candidates of
     * the finger-cube contact in this order, a later one takes over only with a strictly smaller gap: shape3, sph3, shape2, sph2[1],
     * and - only for a cube above upper_check_z, because they hang at the height of the base - sph2[0] and shape1.  shape3 has the axis cap_a -> cap_b and ends (s = 1) in the
     * fingertip sphere: w1 = w2 = rho = cap_radius there.  The joint housings are pucks about the joint axes: one sphere each.   */
    TfLinkShape shape3; TfSphere sph3[1];     /* distal body (lower link + tip link); the joint-3 housing at its top                 */
    TfLinkShape shape2; TfSphere sph2[2];     /* middle link; the joint-2 housing (top) and the joint-3 housing (bottom)             */
    TfLinkShape shape1;                       /* upper link                                                                          */
    float upper_check_z;          /* the upper link and the joint-2 housing are only tested for a cube centre above this height    */
    float middle_check_z;         /* the middle link's body (shape2) is only tested when the highest point of the object is
                                   * above this height (0.075: over the whole joint range the middle link stays >= 0.12 m above the
                                   * floor - tests/test_model_fixture.py -, contact_margin is 0.04)                                  */
    /* cube (cube_multicolor_rrc.urdf:10-18) */
    float cube_half;              /* 0.0325 */
    float cube_mass;              /* 291.3 * 0.065^3 */
    float cube_inertia;           /* m s^2 / 6 (isotropic)                                           */
    float cube_linear_damping, cube_angular_damping;
    /* arena: inner radius of the boundary annulus as a function of height, piecewise linear through the knots (wall_z[i], wall_r[i]):
     * a vertical ring below wall_z[0], the flaring cone of the stage between the knots, nothing above wall_z[3]; from the 40 convex
     * pieces of meshes/convex_table_boundary/convex_*.obj (high_table_boundary.urdf:20-259) via tests/golden/model.npz: mid-way
     * between the chords and the corners of the polygonal inner surface.  The FINGERTIP contact follows the tilt of the surface (normal (c n_h, s) with
     * c, s the cosine and sine of the slope angle of the segment, gap = distance to the tilted surface).  The CUBE CORNERS see the same profile - the radius
     * at the corner's height - with the HORIZONTAL normal at every height: THIS IS THE MODEL, not an omission pending repair.  A cube that lies on the table
     * meets the boundary with its lower corners, below the first knot, where the two normals coincide; a corner on the cone (a lifted or tumbling cube at the
     * boundary: 0.007 % of the env-steps of the bench workload) is stopped radially and gets no vertical impulse where the surface normal would give
     * 0.57 |dv_r| (tests/test_contact_scenarios.py: test_cube_corner_on_the_cone_keeps_the_horizontal_normal, test_random_actions_rarely_put_...). */
    float wall_r[4], wall_z[4];
    /* materials: PhysX "average" combine of trifinger_env.py:364-365,876-878,914-915,934-936 */
    float mu_finger_cube, mu_cube_floor, mu_tip_floor, mu_cube_wall, mu_tip_wall, mu_finger_finger;
    float mu_robot, mu_object, mu_floor, mu_stage;   /* the per-body values behind the averages (1, 1, 0.1, 1): shares of the per-body friction DR */
    float restitution_finger;     /* finger shape vs cube / arena: average(0.8, 0)                   */
    float restitution_ff;         /* finger shape vs finger shape: 0.8                               */
    float bounce_threshold;       /* 0.5 m/s  (scripts/rlg_hydra.py:32)                              */
    float contact_margin;         /* broad phase: a slot is looked at for gaps below this            */
    float contact_slack;          /* a slot gets rows when gap < slack + h max(0, -approach speed)    */
    float contact_offset;         /* 0.002 (scripts/rlg_hydra.py:30): restitution applies below it   */
    float erp;                    /* fraction of penetration removed per substep                     */
    float max_depenetration_velocity;
    float warm_start;             /* fraction of the previous substep's impulses that seeds the solver */
    /* The object as a general box (objects/urdf/cube_multicolor_rrc_phase3.urdf, 20 x 80 x 20 mm, density 500).
     * box == 0: the cube above (isotropic inertia, the fast path); box != 0: half extents and principal moments below,
     * cube_mass = mass of the box, cube_inertia = the reference scalar I_ref of the inertia-scaled solve. */
    int32_t box;
    int32_t box_gyroscopic;       /* free motion includes -w x (I w) (PhysX leaves it out unless asked; default 1 here) */
    float box_half[3];
    float box_inertia[3];
    /* task constants of the object: reference envs/trifinger/utils.py:57-131 (CuboidalObject(size)) */
    float obj_radius_3d;          /* max(size) sqrt(3) / 2                       0.05629165 for the cube */
    float obj_max_com_dist;       /* ARENA_RADIUS (0.195) - radius_3d            0.13870835              */
    float obj_min_height;         /* size_z / 2                                  0.0325                  */
    float obj_span_min_height;    /* max_height (0.1) - min_height               0.0675   (difficulty 3) */
    float obj_span_radius;        /* max_height - radius_3d                      0.04370835 (difficulty 4, 5) */
    /* API 7 / 8.  Finger-finger contacts beyond the three distal pairs: != 0 adds the MIDDLE link of every finger (shape2) against the fingertip capsule
     * of each other finger - six ordered pairs, one frictionless normal row each, visited after the distal pairs in the finger-finger pass (the
     * reference keeps all robot links in one self-colliding group, trifinger_env.py:811-812; uniformly drawn joint positions overlap in a
     * middle-distal pair 1.4 % of the time, in a distal pair 1.7 %).  API 8: tf_default_model sets 1 - the reference's contact set is the default;
     * 0 = the three distal pairs only (`native.ff_middle_pairs: false`, the faster step of API 7).  Cost: INTEGRATION.md. */
    int32_t ff_middle_pairs;
} TfModel;

/* Fill the box fields of `m` for an object of `size` (x, y, z, metres) and `density` (kg/m^3): mass, principal moments,
 * reference inertia, CuboidalObject constants.  A cubic size keeps box = 0 when it equals the default cube. */
void tf_model_set_box(TfModel* m, const float size[3], float density);

typedef struct TfConfig {
    int32_t api_version;          /* TF_API_VERSION */
    int32_t num_envs;             /* envs owned by this handle (this GPU)                            */
    int32_t env_id_offset;        /* global id of local env 0 (RNG is keyed by global id -> results  */
    int32_t global_num_envs;      /*   are invariant to the number of GPUs); total over all ranks    */
    uint64_t seed;
    /* MDP (trifinger_env.py:28-115) */
    int32_t command_mode;
    int32_t normalize_action, normalize_obs, apply_safety_damping;
    int32_t asymmetric_obs, enable_ft_sensors;
    int32_t task_difficulty;
    int32_t episode_length;       /* <= 0: no time-out (None in the reference)                       */
    int32_t control_decimation;
    int32_t robot_reset_type;  float dof_pos_stddev, dof_vel_stddev;
    int32_t object_reset_type;
    int32_t goal_rotation_activate; float goal_rotation_rate_magnitude;
    TfRewardTerm reward[TF_NUM_REWARD_TERMS];
    int32_t finger_reach_norm_p;  /* p of torch.norm in FingerReachObjectRatePenalty (rewards.py:203-235): 1..16, or
                                   * TF_NORM_INF for float('inf'); anything else -> TF_ERR_UNSUPPORTED            */
    float object_rot_scale;       /* ObjectRotationReward.scale  rewards.py:109                      */
    int32_t success_activate; float success_bonus, position_tolerance, orientation_tolerance;
    /* physics stepping (env_base.py:47-70, scripts/rlg_hydra.py:15-35) */
    float dt;                     /* seconds per simulate() call                                     */
    int32_t substeps;             /* solver substeps per simulate()                                  */
    int32_t solver_iterations;    /* num_position_iterations                                         */
    int32_t solver_inner;         /* >= 1 (API 6; default 1).  n > 1: every sweep visits the block of ALL rows that touch the cube (finger-cube, cube-floor,
                                   * cube-boundary, the fingers following through their contact-point velocities) n times before the finger-only rows
                                   * (fingertip-floor, fingertip-boundary, joint limits) get their turn: the block 8 sweeps leave unresolved (INTEGRATION.md
                                   * "what 8 sweeps leave").  Costs what solver_iterations x n sweeps cost. */
    float gravity[3];
    /* Domain randomisation (build-defined: the reference has none, leibnizgym/dr/__init__.py is empty; the intent
     * list is the comment block at trifinger_env.py:385-393).  Scale factors ~ U[lo, hi], drawn per env at reset. */
    int32_t dr_enable;
    float dr_cube_mass[2];
    float dr_cube_size[2];
    float dr_friction[2];
    float dr_motor[2];
    float dr_link_mass[2];        /* one factor for the three moving links of every finger (mass and inertia)      */
    float dr_restitution[2];      /* factor on the finger contact restitution                                      */
    /* the rest of the reference's intent list (trifinger_env.py:387-389): robot base position, stage position, friction of
     * robot / object / stage separately.  Positions: offset ~ U[-a, a] per axis with the half-widths below (metres); the
     * stage is the table with its boundary (object spawn and goal positions move with it).  Friction: one factor ~ U[lo, hi]
     * per body; a contact pair's coefficient (PhysX "average" of the two bodies) is scaled by
     * 1 + s_a (f_a - 1) + s_b (f_b - 1) with s the bodies' shares of the pair's nominal average, on top of dr_friction. */
    float dr_base_pos[3];
    float dr_stage_pos[2];
    float dr_friction_robot[2], dr_friction_object[2], dr_friction_stage[2];
    /* Observation noise (the reference leaves a TODO at trifinger_env.py:979): with dr_enable and dr_obs_noise > 0
     * every step adds dr_obs_noise * U(-1, 1) to the emitted (already scaled) joint positions, joint velocities and
     * object pose of `obs` (slots 0..24); goal, last action and the privileged `states` vector stay exact.  Draws are
     * Philox outputs keyed by (seed, global env id, frame count): reproducible and invariant to the sharding. */
    float dr_obs_noise;
    /* Action repeat (late or dropped command packets): with dr_enable and dr_action_repeat > 0 every env ignores the new
     * action with this probability and re-applies the torque it applied in the previous step (state rows TF_S_TAU; a
     * reset clears them).  `_action_buf` / the obs action slot still report the commanded action.  One Philox draw per
     * env and step, keyed by the frame count at the start of the step. */
    float dr_action_repeat;
    TfModel model;
} TfConfig;

typedef struct TfBuffers {
    float* state;            /* [TF_STATE_ROWS][N]  SoA, resident between steps                      */
    float* action_buf;       /* [N][A] row-major; what the reference calls _action_buf               */
    float* obs;              /* [N][TF_OBS_DIM_BASE + A] row-major                                   */
    float* states;           /* [N][obs_dim + 72] row-major; may be NULL when !asymmetric_obs        */
    float* reward;           /* [N]                                                                  */
    uint8_t* reset_buf;      /* [N] _reset_buf                                                        */
    uint8_t* goal_reset_buf; /* [N] _goal_reset_buf                                                   */
    uint8_t* successes;      /* [N] _successes                                                        */
    uint8_t* dones;          /* [N] reset_buf & goal_reset_buf (env_base.py:399)                      */
    int64_t* steps;          /* [N] _steps_count_buf: int64 as in the reference (torch.long, env_base.py:572); API 5    */
    uint32_t* reset_count;   /* [N] number of RNG draws consumed (Philox counter high word)           */
    float* info;             /* [TF_NUM_INFO]                                                         */
    float* scratch;          /* [tf_scratch_floats(N)] per-wave scratch (developer instrumentation)   */
} TfBuffers;

typedef struct TfHandle_* tf_handle;

int tf_api_version(void);
const char* tf_backend_name(void);              /* "hip-gfx950" or "oracle-c" */
const char* tf_last_error_string(void);
void tf_default_model(TfModel* out);
int tf_action_dim(int32_t command_mode);        /* 9, 9, 18; negative TfStatus on a bad mode */
int64_t tf_scratch_floats(int32_t num_envs);

int tf_create(const TfConfig* cfg, tf_handle* out);
int tf_destroy(tf_handle h);
int tf_bind(tf_handle h, const TfBuffers* bufs);
int tf_set_gravity(tf_handle h, const float g[3]);
/* Fuse the wrapper's clipping (vec_task.py:146-170) into the step: incoming actions are limited to +-clip_actions
 * before anything else sees them (`_action_buf` then holds the limited action, as it does in the reference, where the
 * task receives the clamped tensor), every emitted obs / states value to +-clip_obs (after scaling and after the
 * observation noise).  A bound <= 0 switches that clamp off (the default).  Cold path (blocking parameter copy). */
int tf_set_clipping(tf_handle h, float clip_obs, float clip_actions);
/* The frame count is kept as int64 (reward schedules: env_steps_count = frame count x global_num_envs, in double).  The counter-based draws that
 * are keyed by it - fused action source, observation noise, action repeat - use its LOW 32 BITS as the Philox counter word: their sequences repeat
 * after 2^32 frames (3.5 days at 14 000 steps/s; resets are keyed by the per-env reset count and are not affected). */
int64_t tf_frame_count(tf_handle h);
int tf_set_frame_count(tf_handle h, int64_t frames);
/* The fused step exists in several instantiations with the SAME arithmetic (identical results, bit for bit): a 128-register one that puts four
 * workgroups on a CU (populations that fill the chip) and a 256-register one without spills or LDS parking for populations that never put more
 * than two workgroups on a CU (num_envs <= TF_WIDE_MAX_ENVS: shorter latency per step).  tf_create picks by num_envs (TF_KERNEL_AUTO); the parity
 * tests and the benchmarks force one or the other.  The oracle accepts and ignores the call.  tf_kernel_variant returns what the launches use
 * (TF_KERNEL_NARROW / TF_KERNEL_WIDE / TF_KERNEL_WIDE_HELPERS). */
enum { TF_KERNEL_AUTO = 0, TF_KERNEL_NARROW = 1, TF_KERNEL_WIDE = 2, TF_KERNEL_WIDE_HELPERS = 3 };
#define TF_WIDE_MAX_ENVS 32768
/* TF_KERNEL_WIDE_HELPERS (API 8, additive): the 256-register instantiation in workgroups of EIGHT wavefronts - three helper wavefronts build
 * the middle-distal finger-finger rows (TfModel.ff_middle_pairs) beside the finger wavefronts' contact generation, in the second wavefront slot a CU
 * that holds one workgroup leaves empty -, a fourth runs the distal finger-finger pass in the cube wavefront's place.  Same arithmetic, identical results.
 * TF_KERNEL_AUTO picks it for num_envs <= TF_HELPERS_MAX_ENVS. */
#define TF_HELPERS_MAX_ENVS 16384
int tf_set_kernel_variant(tf_handle h, int32_t variant);
int tf_kernel_variant(tf_handle h);
/* Workgroups of the handle's fused step that fit one CU at once, as the HIP runtime computes it from the kernel's registers and LDS (diagnostic; -1 if
 * the query fails, 0 from the oracle).  The 128-register instantiation is built to fit FOUR (its LDS is 4 x 39.75 KB of the CU's 160 KB, to the
 * byte): anything below that - another toolchain, a runtime that reserves LDS - costs the headline a quarter; tests/test_env_api.py holds it. */
int tf_kernel_occupancy(tf_handle h);

/* The hot path: one control step for every env of the handle, fused, on `stream` (hipStream_t). */
int tf_step(tf_handle h, const float* action /* [N][A] row-major, device */, void* stream);
/* The same step with the action SOURCE fused in: every env draws its own action 2 U[0,1) - 1 per dimension inside the launch
 * (Philox4x32-10 keyed by (seed, global env id, frame count, stream tag): reproducible, invariant to the sharding), exactly what
 * the reference's demo driver feeds the env (scripts/trifinger_random_action.py:33: 2 * torch.rand(...) - 1) and what the
 * BASELINE workload specifies ("random actions 2*U-1").  `action_buf` reports the drawn (clipped) actions as usual. */
int tf_step_random(tf_handle h, void* stream);
/* IsaacEnvBase.reset: reset every env, zero action, ONE simulate, fill obs/states. */
int tf_reset(tf_handle h, void* stream);

/* Optional measurement hook used by bench.py: bracket the fused step kernel (only that kernel) with a pair of
 * events recorded on its launch stream, for up to `max_launches` launches (0 disables).  tf_kernel_time_ms
 * waits for the recorded events and returns the summed kernel duration and the number of launches summed. */
int tf_enable_kernel_timing(tf_handle h, int32_t max_launches);
/* Bracket windows of `window` consecutive launches with one event pair each (default 1): an event pair costs ~3 us of
 * stream time, so bracketing every launch slows the region it measures and reads 2-3 us too long per kernel; over a window
 * of 8 back-to-back launches that cost is amortised.  tf_kernel_time_ms then reports the launches of complete windows. */
int tf_set_kernel_timing_window(tf_handle h, int32_t window);
int tf_kernel_time_ms(tf_handle h, double* total_ms, int64_t* launches);

/* Split path (same arithmetic, one hook per launch) kept for the parity tests. */
int tf_apply_resets(tf_handle h, void* stream);       /* masked _reset_impl then _goal_reset_impl */
int tf_pre_step(tf_handle h, void* stream);            /* action_buf -> state[TF_S_TAU]            */
int tf_simulate(tf_handle h, void* stream);            /* one simulate(): `substeps` solver steps  */
int tf_post_step(tf_handle h, void* stream);           /* obs/states, rewards, termination, info   */
int tf_finish_step(tf_handle h, void* stream);         /* steps += 1, time-out, dones              */

/* Leaf kernels exposed for the golden-vector tests (T7, T11, T12). n rows each. */
int tf_test_quat_diff_rad(const float* a, const float* b, float* out, int32_t n, void* stream);
int tf_test_quat_mul(const float* a, const float* b, float* out, int32_t n, void* stream);
int tf_test_lgsk(const float* x, float scale, float* out, int32_t n, void* stream);
int tf_test_sample_xy(const float* u_radius, const float* u_theta, float r_max, float* x, float* y, int32_t n, void* stream);
int tf_test_sample_yaw_quat(const float* u, float* quat, int32_t n, void* stream);
int tf_test_normalize_quat(const float* normals, float* quat, int32_t n, void* stream);
/* counter-based RNG: 4 uint32 per (seed, env_id, reset_count, stream_id)  */
int tf_test_philox(uint64_t seed, const uint32_t* env_id, const uint32_t* counter, uint32_t stream_id,
                   uint32_t* out4, int32_t n, void* stream);
/* forward kinematics of one finger: q[n][3] -> tip position in the finger base frame [n][3] and the 3x3
 * joint-space mass matrix [n][9] (row-major) + bias forces [n][3] for qd[n][3] */
int tf_test_finger_dynamics(tf_handle h, const float* q, const float* qd, float* tip, float* mass, float* bias,
                            int32_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRIFINGER_H_ */
