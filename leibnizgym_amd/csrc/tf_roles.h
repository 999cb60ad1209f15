// tf_roles.h - the two wavefront roles of the fused TriFinger step.
//
// A workgroup is 256 threads = 4 wavefronts and owns 64 environments, one per lane.  Wavefronts 0..2 are the FINGER ROLE of
// finger 0 / 120 / 240 for those 64 environments, wavefront 3 is the CUBE ROLE (cube, goal, task bookkeeping).  Every
// wavefront is homogeneous (all lanes run the same code on different environments), the four roles of an environment run
// concurrently on the four SIMDs of a CU and meet at workgroup barriers, exchanging a few floats per lane through LDS.
// With 4 workgroups per CU there are 4 wavefronts per SIMD: the latency of one wavefront's dependent instruction stream
// (a lone wavefront issues once per ~5 cycles on a SIMD that can take one instruction per 2) is filled by the others.
//
// The contact solve follows the same split: the finger-cube rows are solved by the cube role in CONTACT SPACE (3x3 block
// A = J M^-1 J^T and contact-point velocity u per finger, published by the finger role), the finger role owns the rows that
// touch only its finger (fingertip-floor, fingertip-wall, joint / velocity limits) and runs them while the cube role runs
// the corner rows (cube-floor, cube-wall): two barriers per sweep.
//
// Every function is the arithmetic twin of its namesake in the test oracle (oracle/tf_oracle.c).
#pragma once
#include <type_traits>

#include "tf_contact.h"

#define NT 256                          // threads per workgroup

// kernel modes: which hooks of the reference step a launch performs (the fused step does all of them)
enum { M_ACT_IN = 1, M_RESETS = 2, M_TORQUE = 4, M_SIM = 8, M_POST = 16, M_FINISH = 32, M_ACT_RAND = 64 /* with M_ACT_IN: the action tile is drawn in the launch */ };
#define M_FUSED_STEP (M_ACT_IN | M_RESETS | M_TORQUE | M_SIM | M_POST | M_FINISH)
#define M_FUSED_STEP_RAND (M_FUSED_STEP | M_ACT_RAND)
#define M_FUSED_RESET (M_RESETS | M_TORQUE | M_SIM | M_POST)

// ---- LDS map: float slots of 64 lanes each, lds[slot * 64 + lane] --------------------------------------------------
// physics phase
#define L_REC(f) (30 * (f))             // contact-space record of finger f
#define R_A 0                           //   6  slots 0..2: K01 K02 K12, the off-diagonal part of the block-local Delassus matrix K = A + D D^T / m + R R^T / I
                                        //      (A = J M^-1 J^T); after the last sweep the six slots carry the friction impulse and the force to the finger role
#define R_DIR 6                         //   9  world directions n, t1, t2
#define R_RXD 15                        //   9  cube arms r x d
#define R_U 24                          //   3  contact-point velocity of the finger side
#define R_DL 27                         //   3  impulse increments of the sweep (cube role -> finger role)
// the same slots before the records exist: what a finger publishes after its free motion (read by the finger-finger pass)
#define P_AW 0
#define P_BW 3
#define P_MINV 6
#define P_S1 12
#define P_C1 13
#define P_P2 14
#define P_P3 17
#define P_VQ 20                         //   -> 23 slots
// cube pose and free velocity published by the cube role for contact generation (free tails of records 0 and 1)
#define L_POSE_A 23                     //   7: cp[3], cq[4]
#define L_POSE_B 53                     //   6: v*[3], w*[3]
#define L_WALL 90                       //  48: corner c at 12 c: r[3], n[2], Dinv[3], bias, lam[3]
// between two substeps: what finger f keeps for the next one, in the first slots of its own record (20 of the 23 slots below L_POSE_A)
#define L_PARK(f) L_REC(f)
#define PK_FC 0                         //   4
#define PK_LINK 4
#define PK_TF 5                         //   3
#define PK_TW 8                         //   3
#define PK_TAU 11                       //   3
#define PK_FT 14                        //   6
#define L_VQFF 138                      //   9: joint velocities after the finger-finger pass; then Dinv[3] of finger f at 3 f
#define L_INIT 147                      //   3: bias of finger f; after the last sweep the normal impulse of finger f
#define L_DR0 32                        //  14: launch prologue only (behind the action tile, before any physics slot is live): the domain-randomisation rows of the
                                        //      env, loaded once by the cube role and handed to the three finger roles through barrier #1
#define LDS_SLOTS 159                   //   (4 workgroups x 159 x 256 B = 159 KB of the CU's 160 KB)
// Mailboxes of the middle-distal finger-finger rows built on the finger wavefronts (the 128-register box kernels keep L_POSE_S at 150..155 and leave
// these rows to the cube role; the 256-register box kernels, which never share a CU with more than one other workgroup, use fresh slots): finger fd receives the velocity change of TWO rows, 2 x 3 floats.  Written between S1 and S1b by the finger that
// owns the middle link, read by fd right behind S1b - so they must be slots nobody writes between S1b and S3 other than fd itself: fd's own L_INIT
// slot, the free tail of its own record (the records are rewritten behind S1b, each by its owner), and the free slots 150..158.
DEV constexpr int ffm_mbox(int fd, int k, bool boxw = false) {   // k = 3 x message + component
    return boxw ? 180 + 6 * fd + k                           //   256-register box kernels (slots 150..155 hold L_POSE_S there): fresh slots 180 .. 197
         : fd == 0 ? (k == 0 ? 147 : 149 + k)               //   finger 0: 147 (L_INIT + 0), 150 .. 154
         : fd == 1 ? (k == 0 ? 148 : (k == 1 ? 59 : 153 + k))    //   finger 1: 148 (L_INIT + 1), 59 (tail of record 1), 155 .. 158
                   : 83 + k;                                 //   finger 2: 83 .. 88 (tail of record 2)
}
// The helper-wavefront instantiation of the 256-register cube kernels (HELP: workgroups of eight wavefronts, one workgroup per CU, populations of at most
// 16384 envs): wavefronts 4..6 build the middle-distal rows of finger 0..2's middle link between S1 and S1b (helper_role) and post BOTH shares - the
// distal finger's through ffm_mbox as above, the owning finger's through ffm_own_mbox; finger f hands them its restitution factor (L_HELP_DR + f).
// Wavefront 7 runs the distal finger-finger pass in the cube role's place, which builds its boundary corners meanwhile.
#define NT_HELP 512
#define L_HELP_OWN 159                  //  18: velocity change of finger fm's own side, 3 x (o - 1) + j at 6 fm
#define L_HELP_DR 177                   //   3: domain-randomisation value 5 (restitution) of the substep, published by finger f with its free motion
#define LDS_SLOTS_HELP 180
DEV constexpr int ffm_own_mbox(int fm, int k) { return L_HELP_OWN + 6 * fm + k; }
#define L_POSE_S 150                    //   6: box kernels only: S = R diag(sqrt(I_ref / I_k)) R^T (00 01 02 11 12 22), published by the cube role
#define LDS_SLOTS_BOX 159
#define LDS_SLOTS_BOX_WIDE 198             //   256-register box kernels (at most two workgroups per CU): + helper slots 159..179, mailboxes 180..197
// post phase (aliases the above)
#define L_XCH (MAX_STATES)              //  18: fingertip position (3) and previous fingertip position (3) of finger f at 6 f
#define L_NAN (MAX_STATES + 18)         //   4: non-finite flag of each role
#define L_VSQ (MAX_STATES + 22)         //   9: squared fingertip speed components of finger f at 3 f (finger_move_penalty: formed where the fingertips are)
#define L_OTERM (MAX_STATES + 31)       //   6: object terms of the reward, parked by the 128-register cube role across P1 / P3 (the 256-register one keeps them in registers)
static_assert(MAX_STATES + 37 <= LDS_SLOTS, "post-phase LDS map");
#define LD(slot) lds[(slot) * WAVE + lane]

struct Ctx {
    int tid, lane, role;
    int wave_first, i, n_valid;
    bool valid;
};

#define ST_RSRC() __builtin_amdgcn_make_buffer_rsrc((void*)P.state, 0, TF_STATE_ROWS * P.N * 4, 0x00020000)
#define LDST(row) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ST_RSRC(), (unsigned)cx.i * 4u, (row) * P.N * 4, 0))
#define STST(row, val) do { if (cx.valid) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(val)), ST_RSRC(), (unsigned)cx.i * 4u, (row) * P.N * 4, 0); } while (0)
#define BAR() __syncthreads()

// Developer instrumentation (libtrifinger_hip_timing.so, tools/phase_timing.py only): lane 0 of every wavefront writes
// s_memtime stamps to scratch[(workgroup * 4 + role) * 64 + id].
#ifdef TF_PHASE_TIMING
#define TF_SCR_STRIDE 256
#define NOW() ((uint32_t)__builtin_readcyclecounter())
#define STAMP(id) do { if (cx.lane == 0) ((gu32*)P.scratch)[((size_t)blockIdx.x * 4 + cx.role) * 64 + (id)] = NOW(); } while (0)
#define STAMPV(id, val) do { if (cx.lane == 0) ((gu32*)P.scratch)[((size_t)blockIdx.x * 4 + cx.role) * 64 + (id)] = (uint32_t)(val); } while (0)
#else
#define TF_SCR_STRIDE 16
#define NOW() 0u
#ifdef TF_PHASE_FENCE      // developer variant: the phase boundaries are scheduling barriers for the compiler (no instruction is emitted)
#define STAMP(id) __builtin_amdgcn_sched_barrier(0)
#else
#define STAMP(id) do { } while (0)
#endif
#define STAMPV(id, val) do { } while (0)
#endif

// ---- cooperative tile moves by the whole workgroup -----------------------------------------------------------------
// store a [n_valid][W] tile staged in LDS as lds[env * W + j] to dst[(wave_first + env) * W + j]: dwordx4, coalesced; the tile
// is a raw buffer of total4 * 16 bytes, so the hardware range check drops the lanes past its end
// NTH: the threads that take part - all four wavefronts (NT), or the three finger wavefronts (NT_F: threads 0..191; the tiles of the post phase,
// which the cube wavefront - the last to finish, it evaluates the rewards - leaves to the fingers)
#define NT_F 192
template <int W, int NTH = NT>
DEV void coop_store_tile(gfloat* __restrict__ dst, const float* lds, const Ctx& cx) {
    const unsigned total = (unsigned)(cx.n_valid * W);
    gfloat* base = dst + (size_t)cx.wave_first * (size_t)W;
    const unsigned total4 = total >> 2;
    const __amdgpu_buffer_rsrc_t tile = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(total4 * 16u), 0x00020000);
    constexpr int ITER = (16 * W + NTH - 1) / NTH;
    const unsigned last4 = total4 - 1u;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const unsigned idx = (unsigned)cx.tid + (unsigned)(NTH * k);
        const unsigned src = (idx < last4) ? idx : last4;
        u32x4 vv = *reinterpret_cast<const u32x4*>(&lds[src * 4u]);
        __builtin_amdgcn_raw_buffer_store_b128(vv, tile, idx * 16u, 0, 0);
    }
    const unsigned tail = (total4 << 2) + (unsigned)cx.tid;
    if (tail < total) base[tail] = lds[tail];
}
// the same for a [n_valid][W] global tile whose rows sit in LDS with row stride LS >= W (obs = first columns of states)
template <int W, int LS, int NTH = NT>
DEV void coop_store_tile_strided(gfloat* __restrict__ dst, const float* lds, const Ctx& cx) {
    const unsigned total = (unsigned)(cx.n_valid * W);
    gfloat* base = dst + (size_t)cx.wave_first * (size_t)W;
    const unsigned total4 = total >> 2;
    const __amdgpu_buffer_rsrc_t tile = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(total4 * 16u), 0x00020000);
    constexpr int ITER = (16 * W + NTH - 1) / NTH;
    const unsigned lastf = total - 1u;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const unsigned idx = (unsigned)cx.tid + (unsigned)(NTH * k);
        u32x4 vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned g = idx * 4u + (unsigned)e;
            g = (g < lastf) ? g : lastf;
            const unsigned row = g / (unsigned)W, col = g - row * (unsigned)W;
            vv[e] = __builtin_bit_cast(unsigned, lds[row * (unsigned)LS + col]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(vv, tile, idx * 16u, 0, 0);
    }
    const unsigned tail = (total4 << 2) + (unsigned)cx.tid;
    if (tail < total) { const unsigned row = tail / (unsigned)W, col = tail - row * (unsigned)W; base[tail] = lds[row * (unsigned)LS + col]; }
}
// load the [n_valid][A] tile src[(wave_first + env) * A + j] into lds[env * A + j] (index clamped instead of branching)
template <int A>
DEV void coop_load_tile(const float* __restrict__ src, float* lds, const Ctx& cx) {
    const float* base = src + (size_t)cx.wave_first * (size_t)A;
    const unsigned last = (unsigned)(cx.n_valid * A) - 1u;
    constexpr int ITER = (WAVE * A + NT - 1) / NT;
    float t[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const unsigned idx = (unsigned)cx.tid + (unsigned)(NT * k);
        t[k] = base[idx < last ? idx : last];
    }
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const unsigned idx = (unsigned)cx.tid + (unsigned)(NT * k);
        if (idx < (unsigned)(WAVE * A)) lds[idx] = t[k];
    }
}

// the [n_valid][A] action tile drawn in place: lds[env * A + c] = 2 u - 1 with u = word (c % 4) of the Philox block
// (global env id, frame count of the step, RNG_ACTION + c / 4).  One block of four values per thread and pass.
template <int A>
DEV void draw_action_tile(const DevParams& P, const StepArgs& sa, float* lds, const Ctx& cx) {
    constexpr int NB = (A + 3) / 4;                   // Philox blocks per env
    for (int t = cx.tid; t < WAVE * NB; t += NT) {
        const int e = t / NB, b = t - e * NB;
        const int ge = cx.wave_first + ((e < cx.n_valid) ? e : (cx.n_valid - 1));
        float u[4];
        rng4(P, (uint32_t)(P.env_id_offset + ge), sa.frame0, RNG_ACTION + (uint32_t)b, u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * b + k;
            if (c < A) lds[e * A + c] = 2.0f * u[k] - 1.0f;
        }
    }
}

// ---- task-layer samplers (reference sample.py) -------------------------------------------------------------------------
DEV void sample_xy(float u_r, float u_t, float r_max, float& x, float& y) {   // sample.py:22-34
    float radius = f_sqrt(u_r) * r_max;
    float s, c;
    tf_sincos(6.2831855f * u_t, s, c);
    x = radius * c;
    y = radius * s;
}
DEV void sample_yaw_quat(float u, float q[4]) {                                 // sample.py:77-84
    float s, c;
    tf_sincos((6.2831855f * u) * 0.5f, s, c);
    q[0] = 0.0f; q[1] = 0.0f; q[2] = s; q[3] = c;
}
DEV void normalize_quat(const float n[4], float q[4]) {                         // sample.py:55-65
    float nrm = f_sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3]);
    float inv = 1.0f / f_max(nrm, 1e-12f);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = n[i] * inv;
}
// the CuboidalObject numbers of the reference (envs/trifinger/utils.py:122-131) come with the model: TfModel.obj_*

// the goal of a reset in two independent halves (trifinger_env.py:1194-1265): position relative to the stage centre + the yaw quaternion of difficulty -1
// (identity otherwise) from the RNG_GOAL_POS block; the orientation of difficulties 4..6 and the angular velocity from their own blocks
DEV void sample_goal_pos(const DevParams& P, uint32_t gid, uint32_t count, float xyz[3], float quat[4]) {
    const int d = P.task_difficulty;
    float u[4];
    rng4(P, gid, count, RNG_GOAL_POS, u);
    float x = 0.0f, y = 0.0f, z;
    quat[0] = 0.0f; quat[1] = 0.0f; quat[2] = 0.0f; quat[3] = 1.0f;
    if (d == -1 || d == 1 || d == 3 || d == 4 || d == 5) sample_xy(u[0], u[1], P.m.obj_max_com_dist, x, y);
    if (d == -1 || d == 1) z = P.m.obj_min_height;
    else if (d == 2 || d == 6) z = P.m.obj_min_height + 0.05f;
    else if (d == 3) z = P.m.obj_span_min_height * u[2] + P.m.obj_min_height;
    else z = P.m.obj_span_radius * u[2] + P.m.obj_radius_3d;
    if (d == -1) sample_yaw_quat(u[3], quat);
    xyz[0] = x; xyz[1] = y; xyz[2] = z;
}
DEV bool goal_has_random_quat(const DevParams& P) { const int d = P.task_difficulty; return d == 4 || d == 5 || d == 6; }
DEV void sample_goal_rot(const DevParams& P, uint32_t gid, uint32_t count, float quat[4], float gw[3]) {      // quat: written for difficulties 4..6 only
    if (goal_has_random_quat(P)) {
        float v[4], n[4];
        rng4(P, gid, count, RNG_GOAL_QUAT, v);
        box_muller(v[0], v[1], n[0], n[1]);
        box_muller(v[2], v[3], n[2], n[3]);
        normalize_quat(n, quat);
    }
    if (P.goal_rotation_activate) {
        float v[4], n[4];
        rng4(P, gid, count, RNG_GOAL_ANGVEL, v);
        box_muller(v[0], v[1], n[0], n[1]);
        box_muller(v[2], v[3], n[2], n[3]);
        float nrm = f_sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        float mag = n[3] * P.goal_rate;
#pragma unroll
        for (int i = 0; i < 3; ++i) gw[i] = mag * (n[i] / nrm);
    } else {
        gw[0] = 0.0f; gw[1] = 0.0f; gw[2] = 0.0f;
    }
}
// raw[0..2] position relative to the stage centre, raw[3..6] orientation, raw[7..9] angular velocity -> the goal rows (goals move with the stage)
template <bool EXT>
DEV void goal_from_raw(const float raw[10], const float dr[TF_NUM_DR], float gp[3], float gq[4], float gw[3]) {
    gp[0] = EXT ? raw[0] + dr[TF_DR_STAGE_POS] : raw[0]; gp[1] = EXT ? raw[1] + dr[TF_DR_STAGE_POS + 1] : raw[1]; gp[2] = raw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) gq[i] = raw[3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) gw[i] = raw[7 + i];
}
template <bool EXT>
DEV void sample_goal(const DevParams& P, uint32_t gid, uint32_t count, const float dr[TF_NUM_DR], float gp[3], float gq[4], float gw[3]) {
    float raw[10];
    sample_goal_pos(P, gid, count, &raw[0], &raw[3]);
    sample_goal_rot(P, gid, count, &raw[3], &raw[7]);
    goal_from_raw<EXT>(raw, dr, gp, gq, gw);
}

// per-env domain-randomisation factors drawn at a reset (build-defined): scale = lo + (hi - lo) u
template <bool EXT>
DEV void draw_dr(const DevParams& P, uint32_t gid, uint32_t count, float dr[TF_NUM_DR]) {
    float u[4];
    rng4(P, gid, count, RNG_DR, u);
    dr[0] = FMA(P.dr_cube_mass[1] - P.dr_cube_mass[0], u[0], P.dr_cube_mass[0]);
    dr[1] = FMA(P.dr_cube_size[1] - P.dr_cube_size[0], u[1], P.dr_cube_size[0]);
    dr[2] = FMA(P.dr_friction[1] - P.dr_friction[0], u[2], P.dr_friction[0]);
    dr[3] = FMA(P.dr_motor[1] - P.dr_motor[0], u[3], P.dr_motor[0]);
    rng4(P, gid, count, RNG_DR + 1u, u);
    dr[4] = FMA(P.dr_link_mass[1] - P.dr_link_mass[0], u[0], P.dr_link_mass[0]);
    dr[5] = FMA(P.dr_restitution[1] - P.dr_restitution[0], u[1], P.dr_restitution[0]);
    if (!EXT) return;
    // robot base and stage positions: offset = a (2 u - 1) per axis; friction per body
    rng4(P, gid, count, RNG_DR + 2u, u);
#pragma unroll
    for (int k = 0; k < 3; ++k) dr[TF_DR_BASE_POS + k] = P.dr_base_pos[k] * (2.0f * u[k] - 1.0f);
    dr[TF_DR_STAGE_POS] = P.dr_stage_pos[0] * (2.0f * u[3] - 1.0f);
    rng4(P, gid, count, RNG_DR + 3u, u);
    dr[TF_DR_STAGE_POS + 1] = P.dr_stage_pos[1] * (2.0f * u[0] - 1.0f);
    dr[TF_DR_FRICTION_ROBOT] = FMA(P.dr_friction_robot[1] - P.dr_friction_robot[0], u[1], P.dr_friction_robot[0]);
    dr[TF_DR_FRICTION_OBJECT] = FMA(P.dr_friction_object[1] - P.dr_friction_object[0], u[2], P.dr_friction_object[0]);
    dr[TF_DR_FRICTION_STAGE] = FMA(P.dr_friction_stage[1] - P.dr_friction_stage[0], u[3], P.dr_friction_stage[0]);
}
// 1 + s_a (f_a - 1) + s_b (f_b - 1) with the shares s = mu / (mu_a + mu_b) of the two bodies in the pair's average
DEV float pair_factor(float mu_a, float fa1, float mu_b, float fb1) {
    const float inv = 1.0f / (mu_a + mu_b);
    return FMA(mu_b * inv, fb1, FMA(mu_a * inv, fa1, 1.0f));
}

// ---- observation emission: scale_transform (reference torch_utils.py:18-36) as one FMA per slot, then the fused wrapper
// clipping.  Every slot but the action one has limits that are constants of the MDP (trifinger_env.py:153-213). ----------
DEV float emit_scaled(float x, float lo, float hi, bool nrm, float co) {
    const float o_ = (lo + hi) * 0.5f, i_ = 1.0f / (hi - lo);
    return f_clamp(nrm ? FMA(x, 2.0f * i_, -(2.0f * o_) * i_) : x, -co, co);
}
DEV float emit_table(const DevParams& P, int col, float x, bool nrm, float co) {
    const float off = P.tables[TAB_OFF + col], inv = P.tables[TAB_INV + col];
    return f_clamp(nrm ? FMA(x, 2.0f * inv, -(2.0f * off) * inv) : x, -co, co);
}
#define QLO(j) (((j) % 3 == 0) ? -0.33f : (((j) % 3 == 1) ? 0.0f : -2.7f))
#define QHI(j) (((j) % 3 == 0) ? 1.0f : (((j) % 3 == 1) ? 1.57f : 0.0f))
#define PLO(j) (((j) == 2) ? 0.0f : -0.3f)
#define TLO(j) (((j) < 2) ? -0.4f : (((j) == 2) ? 0.0f : (((j) < 7) ? -1.0f : -0.2f)))
#define THI(j) (((j) < 2) ? 0.4f : (((j) == 2) ? 0.5f : (((j) < 7) ? 1.0f : 0.2f)))

DEV float norm3d(const float a[3], const float b[3]) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return f_sqrt(dx * dx + dy * dy + dz * dz);
}
// torch.norm(a - b, p) for the finger_reach_object_rate term (reference rewards.py:216-226): p = 2 is the Euclidean norm above;
// p = 1 and the maximum norm are exact; an integer p in 3..16 takes the p-th root as exp(log(s) / p) followed by one Newton
// step on y^p = s, which brings the polynomial exp / log (1e-7 relative) down to rounding error
DEV float ipow(float x, int n) {
    float t = x;
    for (int k = 1; k < n; ++k) t = t * x;
    return t;
}
DEV float norm_p3(const float a[3], const float b[3], int p) {
    if (__builtin_expect(p == 2, 1)) return norm3d(a, b);
    const float ax = f_abs(a[0] - b[0]), ay = f_abs(a[1] - b[1]), az = f_abs(a[2] - b[2]);
    if (p == 1) return (ax + ay) + az;
    if (p == TF_NORM_INF) return f_max(f_max(ax, ay), az);
    // the largest component is taken out first: d^p of a 4 mm distance underflows fp32 from p = 10 on
    const float mx = f_max(f_max(ax, ay), az);
    if (!(mx > 0.0f)) return 0.0f;
    const float s = (ipow(ax / mx, p) + ipow(ay / mx, p)) + ipow(az / mx, p);      // in [1, 3]
    float y = tf_exp(tf_log(s) / (float)p);
    const float yp1 = ipow(y, p - 1);
    y = y - (yp1 * y - s) / ((float)p * yp1);
    return mx * y;
}

// fingertip link state in the world frame: position, quaternion (xyzw), linear and angular velocity
DEV void tip_state(const TfModel& m, const Yaw& y, const FK& k, const float q[3], const float qd[3], float out[13]) {
    float To[3];
    link_point<3>(k, m.tip_origin, To);
    base_to_world(y, To, &out[0]);
    float sy, cy, sx, cxx;
    tf_sincos(0.5f * q[0], sy, cy);
    tf_sincos(0.5f * (q[1] + q[2]), sx, cxx);
    float qyx[4] = {cy * sx, sy * cxx, -(sy * sx), cy * cxx};
    float qz[4] = {0.0f, 0.0f, y.hs, y.hc};
    quat_mul(qz, qyx, &out[3]);
    float L1[3], L2[3], L3[3], vb[3], wb[3];
    levers(k, To, L1, L2, L3);
#pragma unroll
    for (int i = 0; i < 3; ++i) vb[i] = L1[i] * qd[0] + L2[i] * qd[1] + L3[i] * qd[2];
    wb[0] = k.ax[0] * qd[1] + k.ax[0] * qd[2];
    wb[1] = qd[0];
    wb[2] = k.ax[2] * qd[1] + k.ax[2] * qd[2];
    dir_base_to_world(y, vb, &out[7]);
    dir_base_to_world(y, wb, &out[10]);
}

// =====================================================================================================================
// FINGER ROLE
// =====================================================================================================================
// what a finger publishes after its free motion, as the registers of a reader (the finger-finger rows)
struct FingerPubRegs { FK k; };
DEV void read_pub(const float* lds, int lane, int f, FingerPubRegs& p) {
    const int pb = L_REC(f);
#pragma unroll
    for (int j = 0; j < 3; ++j) { p.k.p2[j] = LD(pb + P_P2 + j); p.k.p3[j] = LD(pb + P_P3 + j); }
#pragma unroll
    for (int j = 0; j < 6; ++j) p.k.Minv[j] = LD(pb + P_MINV + j);
    p.k.s1 = LD(pb + P_S1); p.k.c1 = LD(pb + P_C1);
    p.k.ax[0] = p.k.c1; p.k.ax[1] = 0.0f; p.k.ax[2] = -p.k.s1;
}

struct TipContact {            // fingertip sphere against one feature of the arena: finger-only rows
    bool active;
    float J[9], Dinv[3], bias, lam[3], mu;
    float arm[3];             // contact point relative to the tip-link origin (fingertip wrench sensor)
};

template <int A, bool IS_RESET, bool ASYM, int MODE, int X, bool WIDE, bool HELP = false>
DEV void finger_role(const DevParams& P, const StepArgs& sa, const float* __restrict__ action, float* lds, const Ctx& cx) {
    static_assert(!HELP || WIDE, "helper wavefronts: 256-register kernels only");
    constexpr bool EXT = X != 0;      // X: 0 the headline kernels, 1 extended domain randomisation, 2 the same with the general box object
    // WIDE: the 256-register instantiation (2 wavefronts per SIMD) launched for populations that never put more than two workgroups on a
    // CU (num_envs <= 32768): nothing is parked in LDS or re-read from the state rows between substeps.  Same arithmetic, bit for bit.
    constexpr bool BOXK = X == 2;
    const TfModel& m = P.m;
    const int f = cx.role, lane = cx.lane;
    constexpr int OD = TF_OBS_DIM_BASE + A, SD = OD + TF_STATES_EXTRA;
    constexpr int TW = ASYM ? SD : OD;                 // width of the tile staged in LDS in the post phase
    constexpr int NDR = EXT ? TF_NUM_DR : TF_DR_BASE_POS;   // the base / stage / per-body friction slots exist in the EXT kernels only
    constexpr int AJ = A / 3;                          // action values of one finger: 3, or 3 + 3 stiffnesses
    const Yaw yw = {m.base_yaw_cos[f], m.base_yaw_sin[f], m.base_half_yaw_cos[f], m.base_half_yaw_sin[f], m.base_height};
    const uint32_t gid = (uint32_t)(P.env_id_offset + cx.i);
    // ---- loads ----
    float q[3], qd[3], tau[3], dr[TF_NUM_DR];
    uint8_t fl_reset = 0;
    uint32_t fl_count = 0;
    STAMP(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) { q[j] = LDST(TF_S_Q + 3 * f + j); qd[j] = LDST(TF_S_QD + 3 * f + j); }
    // (the domain-randomisation rows of the env are loaded once per workgroup, by the cube role, and arrive through LDS behind barrier #1: four
    // wavefronts fetching the same 14 rows more than doubled the load burst every workgroup of a launch starts with)
    if (!(MODE & M_TORQUE) && (MODE & (M_RESETS | M_SIM))) {
#pragma unroll
        for (int j = 0; j < 3; ++j) tau[j] = LDST(TF_S_TAU + 3 * f + j);
    }
    float tau_prev[3] = {0.0f, 0.0f, 0.0f};                      // action repeat: the torque of the previous step, requested with the other loads of the prologue
    if ((MODE & M_TORQUE) && !IS_RESET && P.dr_action_repeat > 0.0f) {
#pragma unroll
        for (int j = 0; j < 3; ++j) tau_prev[j] = LDST(TF_S_TAU + 3 * f + j);
    }
    // activity code of this finger's warm-start rows (row TF_S_FC_LINK): link that held the finger-cube contact + 4 if the fingertip-wall
    // contact pushed.  The rows of an inactive contact are neither loaded nor stored (their content is then undefined).
    float fc_code = 0.0f;
    if (MODE & M_SIM) fc_code = LDST(TF_S_FC_LINK + f);
    if (MODE & M_RESETS) { fl_reset = P.reset_buf[(unsigned)cx.i]; fl_count = P.reset_count[(unsigned)cx.i]; }
    // the fused step draws the samples of an env's NEXT reset at its end (last block of this role): what it needs to know which envs will be flagged
    constexpr bool PRESAMPLE = !IS_RESET && (MODE & M_RESETS) && (MODE & M_SIM) && (MODE & M_POST) && (MODE & M_FINISH);
    int fl_steps = 0;
    uint8_t fl_goal_reset = 0;
    if (PRESAMPLE) { fl_steps = (int)P.steps[(unsigned)cx.i]; fl_goal_reset = P.goal_reset_buf[(unsigned)cx.i]; }
    if (MODE & M_ACT_RAND) draw_action_tile<A>(P, sa, lds, cx);
    else if (MODE & M_ACT_IN) coop_load_tile<A>(action, lds, cx);
    else if (MODE & (M_RESETS | M_TORQUE | M_POST)) coop_load_tile<A>((const float*)P.action_buf, lds, cx);
    BAR();                                                      // #1: action tile in LDS, flag loads have returned
#pragma unroll
    for (int j = 0; j < TF_NUM_DR; ++j) dr[j] = (j < NDR && P.dr_enable) ? LD(L_DR0 + j) : TF_DR_NEUTRAL(j);
    // L_DR0 aliases finger 1's record: in the fused modes barriers #2a / #2b stand between these reads and the first physics write; a launch
    // without them (the split path's tf_simulate / tf_post_step) closes the hand-over with a barrier of its own
    if (!(MODE & (M_ACT_IN | M_RESETS))) BAR();                 // #1b
    STAMP(1);
    // The warm-start rows are first needed when the contact rows are built, a free-motion phase later: issued here, behind the
    // barrier, they stay out of the load burst every workgroup of the launch starts with.
    // (the warm-start rows themselves are loaded in the first substep, where the contact rows are built: kept across the free motion
    // they would only be spilled)
    // ---- masked _reset_impl for this finger (trifinger_env.py:373-423, 1101-1147) ----
    const bool rflag = (MODE & M_RESETS) && (IS_RESET || fl_reset != 0);
    if (MODE & M_RESETS) {
        if (rflag) {
            if (P.dr_enable) draw_dr<EXT>(P, gid, fl_count, dr);
            if (P.robot_reset_type == TF_RESET_DEFAULT) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { q[j] = m.q_default[j]; qd[j] = 0.0f; }
            } else if (P.robot_reset_type == TF_RESET_RANDOM) {
                float n[20];
#pragma unroll
                for (int b = 0; b < 5; ++b) rng4(P, gid, fl_count, RNG_ROBOT + (uint32_t)b, &n[4 * b]);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float nq = 0.0f, nv = 0.0f;
#pragma unroll
                    for (int jj = 0; jj < 9; ++jj) { nq = (jj == 3 * f + j) ? n[jj] : nq; nv = (jj == 3 * f + j) ? n[9 + jj] : nv; }
                    q[j] = m.q_default[j] + P.dof_pos_stddev * (2.0f * nq - 1.0f);
                    qd[j] = 0.0f + P.dof_vel_stddev * (2.0f * nv - 1.0f);
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) tau[j] = 0.0f;          // the stored torque (what an action repeat re-applies)
            if (MODE & M_SIM) fc_code = -1.0f;                  // ... and the solver warm start (-1: the rows are not looked at)
        }
    }
    // ---- this finger's action values: clipped, zeroed by a reset (trifinger_env.py:387), written back for _action_buf ----
    float act[AJ];
    if (MODE & (M_ACT_IN | M_RESETS | M_TORQUE | M_POST)) {
        const int row = cx.valid ? lane : (cx.n_valid - 1);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int col = (j < 3) ? (3 * f + j) : (9 + 3 * f + (j - 3));
            float a = lds[row * A + col];
            if (MODE & (M_ACT_IN | M_RESETS)) a = f_clamp(a, -P.clip_act, P.clip_act);
            if (IS_RESET || rflag) a = 0.0f;
            act[j] = a;
        }
        if (MODE & (M_ACT_IN | M_RESETS)) {
            BAR();                                              // #2a: every lane has read its row (rows of invalid lanes alias the last one)
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int col = (j < 3) ? (3 * f + j) : (9 + 3 * f + (j - 3));
                if (cx.valid) lds[lane * A + col] = act[j];
            }
            BAR();                                              // #2b
            coop_store_tile<A>(P.action_buf, lds, cx);
        }
    }
    // ---- _pre_step torque law for the three joints of this finger (trifinger_env.py:442-494) ----
    if (MODE & M_TORQUE) {
        float t[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int jg = 3 * f + j;
            float at = act[j], ak = (AJ == 6) ? act[3 + (j % 3)] : 0.0f;
            if (P.normalize_action) {
                float lo = P.tables[TAB_ACT_LO + jg], hi = P.tables[TAB_ACT_HI + jg];
                float off = (lo + hi) * 0.5f;
                at = at * (hi - lo) * 0.5f + off;
                if (AJ == 6) {
                    float lo2 = P.tables[TAB_ACT_LO + 9 + jg], hi2 = P.tables[TAB_ACT_HI + 9 + jg];
                    float off2 = (lo2 + hi2) * 0.5f;
                    ak = ak * (hi2 - lo2) * 0.5f + off2;
                }
            }
            float tq;
            if (P.command_mode == TF_CMD_TORQUE) tq = at;
            else if (P.command_mode == TF_CMD_POSITION) { tq = P.tables[TAB_KP + jg] * (at - q[j]); tq = tq - P.tables[TAB_KD + jg] * qd[j]; }
            else { tq = ((AJ == 6) ? ak : at) * (at - q[j]); tq = tq - P.tables[TAB_KD + jg] * qd[j]; }
            tq = f_max(f_min(tq, 0.36f), -0.36f);
            if (P.apply_safety_damping) {
                tq = tq - P.tables[TAB_KS + jg] * qd[j];
                tq = f_max(f_min(tq, 0.36f), -0.36f);
            }
            t[j] = tq * dr[3];                                  // domain randomisation of the motor strength (1.0 when off)
        }
        if (!IS_RESET && P.dr_action_repeat > 0.0f) {           // build-defined action repeat: keep the previous step's torque
            float u[4];
            rng4(P, gid, sa.frame0, RNG_ACT_REPEAT, u);
            const bool keep = u[0] < P.dr_action_repeat;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float prev = rflag ? 0.0f : tau_prev[j];
                t[j] = keep ? prev : t[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) tau[j] = t[j];
        if (!(MODE & M_SIM)) {                                  // fingertip wrench accumulator of the step: a launch that also simulates
#pragma unroll                                                  // starts from zero in registers (and the fused step never stores it)
            for (int j = 0; j < 6; ++j) STST(TF_S_FT + 6 * f + j, 0.0f);
        }
    }
    if (MODE & (M_TORQUE | M_RESETS)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) STST(TF_S_TAU + 3 * f + j, tau[j]);
    }
    if (MODE & M_RESETS) {                                      // a reset clears the solver warm start: activity code 0 (the finger-cube and
        if (rflag) {                                            // fingertip-wall rows are then ignored), fingertip-floor rows zero
            STST(TF_S_FC_LINK + f, 0.0f);
#pragma unroll
            for (int j = 0; j < 3; ++j) STST(TF_S_LAM_TF + 3 * f + j, 0.0f);
        }
    }
    // =================================================================================================================
    // physics: decimation x substeps solver substeps
    // =================================================================================================================
    STAMP(2);
    // A launch that simulates AND emits the observations (the fused step) hands the fingertip wrench of the step from its last substep to
    // the post phase in registers: the TF_S_FT rows ("split path only") are neither written nor read by it.
    constexpr bool FT_IN_REGS = ((MODE & M_SIM) != 0) && ((MODE & M_POST) != 0);      // (in LDS, to be exact: the parking slots of the last substep)
    float k_ft[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};      // WIDE: the fingertip wrench accumulator, carried in registers through the substeps into the post phase
    if (MODE & M_SIM) {
        const float h = P.hsub, inv_h = 1.0f / h;
        const int nsub = sa.nsim * P.substeps;
        float k_lam_fc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, k_fc_link = 0.0f, k_lam_tf[3] = {0.0f, 0.0f, 0.0f}, k_lam_tw[3] = {0.0f, 0.0f, 0.0f};   // WIDE: carried in registers between substeps
        for (int s = 0; s < nsub; ++s) {
            const int sb_ = 4 + 12 * (s & 1);
            (void)sb_;
            // Values that are cold through the sweeps (torque, last substep's impulses, wrench accumulator) do not occupy registers the
            // sweeps need: between two substeps of a launch they are parked in the first slots of this finger's own record in LDS
            // (L_PARK: dead from the last sweep of a substep until the finger publishes its free motion in the next one; nobody else
            // writes slots 0..22 of a record in that window); the state rows are read at the first substep and written at the last.
            float drs[TF_NUM_DR], taus[3];
            float lam_fc[4], fc_link, lam_tf[3], lam_tw[3];     // impulses of the last substep (issued here, used in F2)
            float ft_run[6];                                    // fingertip wrench accumulator of the step (its state rows)
            if (s == 0) {                                       // first substep: what the step started with is still in registers
#pragma unroll
                for (int j = 0; j < TF_NUM_DR; ++j) drs[j] = dr[j];
#pragma unroll
                for (int j = 0; j < 3; ++j) { taus[j] = tau[j]; lam_tf[j] = 0.0f; lam_tw[j] = 0.0f; }
#pragma unroll
                for (int j = 0; j < 4; ++j) lam_fc[j] = 0.0f;
                fc_link = 0.0f;
                if (ASYM) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) ft_run[j] = (MODE & M_TORQUE) ? 0.0f : LDST(TF_S_FT + 6 * f + j);
                }
            } else if (WIDE) {
#pragma unroll
                for (int j = 0; j < TF_NUM_DR; ++j) drs[j] = dr[j];
#pragma unroll
                for (int j = 0; j < 3; ++j) { taus[j] = tau[j]; lam_tf[j] = k_lam_tf[j]; lam_tw[j] = k_lam_tw[j]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) lam_fc[j] = k_lam_fc[j];
                fc_link = k_fc_link;
                if (ASYM) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) ft_run[j] = k_ft[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < TF_NUM_DR; ++j) drs[j] = (j < NDR && P.dr_enable) ? LDST(TF_S_DR + j) : TF_DR_NEUTRAL(j);
#pragma unroll
                for (int j = 0; j < 3; ++j) taus[j] = LD(L_PARK(f) + PK_TAU + j);
#pragma unroll
                for (int j = 0; j < 4; ++j) lam_fc[j] = LD(L_PARK(f) + PK_FC + j);
                fc_link = LD(L_PARK(f) + PK_LINK);
#pragma unroll
                for (int j = 0; j < 3; ++j) { lam_tf[j] = LD(L_PARK(f) + PK_TF + j); lam_tw[j] = LD(L_PARK(f) + PK_TW + j); }
                if (ASYM) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) ft_run[j] = LD(L_PARK(f) + PK_FT + j);
                }
            }
            const float* dr = drs;
            const float* tau = taus;
            const float cube_mass = m.cube_mass * dr[0];
            const float cube_inertia = m.cube_inertia * dr[0] * dr[1] * dr[1];
            const float inv_m = 1.0f / cube_mass, inv_I = 1.0f / cube_inertia;
            // per-body friction factors, robot base and stage offsets (neutral constants when the feature is off)
            const float fr1 = dr[TF_DR_FRICTION_ROBOT] - 1.0f, fo1 = dr[TF_DR_FRICTION_OBJECT] - 1.0f, fs1 = dr[TF_DR_FRICTION_STAGE] - 1.0f;
            const float mu_fc = (m.mu_finger_cube * dr[2]) * (EXT ? pair_factor(m.mu_robot, fr1, m.mu_object, fo1) : 1.0f);
            const float mu_tf = (m.mu_tip_floor * dr[2]) * (EXT ? pair_factor(m.mu_robot, fr1, m.mu_floor, fs1) : 1.0f);
            const float mu_tw = (m.mu_tip_wall * dr[2]) * (EXT ? pair_factor(m.mu_robot, fr1, m.mu_stage, fs1) : 1.0f);
            const float rest_f = m.restitution_finger * dr[5];
            const float boff[3] = {dr[TF_DR_BASE_POS], dr[TF_DR_BASE_POS + 1], dr[TF_DR_BASE_POS + 2]};
            const float soff[2] = {dr[TF_DR_STAGE_POS], dr[TF_DR_STAGE_POS + 1]};
            const float ws = m.warm_start;
            constexpr bool box = BOXK;                        // general box object: its own kernel instantiations, the cube kernels carry none of it
            float hc[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) hc[j] = (box ? m.box_half[j] : m.cube_half) * dr[1];
            // ---- F1: free motion ----
            FK k;
            float vq[3], Aw[3], Bw[3], Tw[3];
            {
                float M[6], bias[3], rhs[3], acc[3];
                fk_setup(m, q, k);
                finger_dynamics(m, k, qd, P.grav, M, bias);
#pragma unroll
                for (int j = 0; j < 6; ++j) M[j] = M[j] * dr[4];
#pragma unroll
                for (int j = 0; j < 3; ++j) bias[j] = bias[j] * dr[4];
                inv3sym(M, k.Minv);
#pragma unroll
                for (int j = 0; j < 3; ++j) rhs[j] = tau[j] - bias[j];
                sym3_mul(k.Minv, rhs, acc);
                const float damp = 1.0f - h * m.link_angular_damping;
#pragma unroll
                for (int j = 0; j < 3; ++j) vq[j] = FMA(h, acc[j], qd[j]) * damp;
                float Ab[3], Bb[3], To[3];
                link_point<3>(k, m.cap_a, Ab);
                link_point<3>(k, m.cap_b, Bb);
                link_point<3>(k, m.tip_origin, To);
                base_to_world(yw, Ab, Aw);
                base_to_world(yw, Bb, Bw);
                base_to_world(yw, To, Tw);
                const int pb = L_REC(f);
#pragma unroll
                for (int j = 0; j < 3; ++j) { LD(pb + P_AW + j) = Aw[j]; LD(pb + P_BW + j) = Bw[j]; LD(pb + P_P2 + j) = k.p2[j]; LD(pb + P_P3 + j) = k.p3[j]; LD(pb + P_VQ + j) = vq[j]; }
#pragma unroll
                for (int j = 0; j < 6; ++j) LD(pb + P_MINV + j) = k.Minv[j];
                LD(pb + P_S1) = k.s1; LD(pb + P_C1) = k.c1;
                if (HELP) LD(L_HELP_DR + f) = dr[5];
            }
            STAMP(sb_ + 0);
            BAR();                                              // S1: free motion of every role published
            STAMP(sb_ + 1);
            // ---- F2: contact generation (positions at the start of the substep) ----
            float cp[3], cq[4], R[9], cvf[3], cwf[3];             // cube pose and free velocity (their slots are reused after S1b)
#pragma unroll
            for (int j = 0; j < 3; ++j) { cp[j] = EXT ? LD(L_POSE_A + j) - boff[j] : LD(L_POSE_A + j); cvf[j] = LD(L_POSE_B + j); cwf[j] = LD(L_POSE_B + 3 + j); }   // robot frame = world - base offset
#pragma unroll
            for (int j = 0; j < 4; ++j) cq[j] = LD(L_POSE_A + 3 + j);
            quat_to_rot(cq, R);
            if (s == 0 && !(fc_code < 0.0f)) {                  // warm-start rows of the step (wave-uniform branches; consumed at the end of this phase)
                const int code = (int)fc_code;
                const bool fc_was = (code & 3) != 0, tw_was = (code & 4) != 0;
                fc_link = (float)(code & 3);
#pragma unroll
                for (int j = 0; j < 3; ++j) lam_tf[j] = LDST(TF_S_LAM_TF + 3 * f + j);
                if (__builtin_amdgcn_ballot_w64(fc_was) != 0ull) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float t = LDST(TF_S_LAM_FC + 4 * f + j); lam_fc[j] = fc_was ? t : 0.0f; }
                }
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(tw_was) != 0ull, 0)) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) { const float t = LDST(TF_S_LAM_TW + 3 * f + j); lam_tw[j] = tw_was ? t : 0.0f; }
                }
            }
            // finger vs cube: the shape with the smallest gap holds the contact (TfLinkShape, include/trifinger.h): the distal body
            // first - its axis is the fingertip capsule's, it ends in the fingertip sphere -, its housing sphere, the middle link with
            // its joint-3 housing sphere and - only for a cube above upper_check_z, practically never - its joint-2 housing sphere and the
            // upper link; a later candidate
            // takes over only with a strictly smaller gap.  All shape constants are scalars of the parameter block.
            float gap = 0.0f, x[3], y[3], nc[3], radius = 0.0f;
            int link = 0;
            // a point of link LK (link frame) in the cube frame
            auto to_cube = [&](auto lkc, const float local[3], float out[3]) __attribute__((always_inline)) {
                constexpr int LK = decltype(lkc)::value;
                float Pb_[3], Pw_[3];
                link_point<LK>(k, local, Pb_);
                base_to_world(yw, Pb_, Pw_);
                float dd[3] = {Pw_[0] - cp[0], Pw_[1] - cp[1], Pw_[2] - cp[2]};
                mat3T_mul(R, dd, out);
            };
            // tapered rounded box of link LK against the cube: closest points of its axis (a, b: cube frame), then the support function
            // of the cross-section at that point of the axis along the direction to the cube
            auto try_shape = [&](auto lkc, const TfLinkShape& sh, const float a[3], const float b[3], bool allowed) __attribute__((always_inline)) {
                constexpr int LK = decltype(lkc)::value;
                float gx[3], gy[3], gn[3], D, sp;
                seg_box(a, b, hc, 0.0f, D, gx, gy, gn, sp);
                float uw[3], ub[3], ul[3];
                mat3_mul(R, gn, uw);                            // world direction from the cube point to the axis point
                dir_world_to_base(yw, uw, ub);
                rot_link_T<LK>(k, ub, ul);
                const float u1 = -ul[0], u2 = (LK == 1) ? -ul[2] : -ul[1];      // towards the cube, along the two width directions
                const float rho = FMA(sp, sh.rho[1] - sh.rho[0], sh.rho[0]);
                const float h1 = FMA(sp, sh.w1[1] - sh.w1[0], sh.w1[0]) - rho, h2 = FMA(sp, sh.w2[1] - sh.w2[0], sh.w2[0]) - rho;
                const float o1 = FMA(sp, sh.o1[1] - sh.o1[0], sh.o1[0]), o2 = FMA(sp, sh.o2[1] - sh.o2[0], sh.o2[0]);
                const float ext = FMA(o2, u2, FMA(o1, u1, FMA(h2, f_abs(u2), FMA(h1, f_abs(u1), rho))));
                const float gg = D - ext;
                const bool take = allowed && ((link == 0) || (gg < gap));
                link = take ? LK : link; gap = take ? gg : gap; radius = take ? ext : radius;
#pragma unroll
                for (int j = 0; j < 3; ++j) { x[j] = take ? gx[j] : x[j]; y[j] = take ? gy[j] : y[j]; nc[j] = take ? gn[j] : nc[j]; }
            };
            auto try_sphere = [&](auto lkc, const TfSphere& sp, bool allowed) __attribute__((always_inline)) {
                constexpr int LK = decltype(lkc)::value;
                float c[3], gy[3], gn[3], gg;
                to_cube(lkc, sp.c, c);
                point_box(c, hc, sp.radius, gg, gy, gn);
                const bool take = allowed && (gg < gap);
                link = take ? LK : link; gap = take ? gg : gap; radius = take ? sp.radius : radius;
#pragma unroll
                for (int j = 0; j < 3; ++j) { x[j] = take ? c[j] : x[j]; y[j] = take ? gy[j] : y[j]; nc[j] = take ? gn[j] : nc[j]; }
            };
            {
                using L3 = std::integral_constant<int, 3>; using L2 = std::integral_constant<int, 2>; using L1 = std::integral_constant<int, 1>;
                {   // distal body: its axis end points are the fingertip capsule's (Aw, Bw of the free-motion phase)
                    float da[3] = {Aw[0] - cp[0], Aw[1] - cp[1], Aw[2] - cp[2]};
                    float db[3] = {Bw[0] - cp[0], Bw[1] - cp[1], Bw[2] - cp[2]};
                    float a[3], b[3];
                    mat3T_mul(R, da, a);
                    mat3T_mul(R, db, b);
                    try_shape(L3{}, m.shape3, a, b, true);
                }
                try_sphere(L3{}, m.sph3[0], true);
                // the middle link never comes lower than 0.12 m: it can only matter for an object whose highest
                // point is above middle_check_z - a lifted or tumbling one (wave-level branch: a fifth of the wavefronts under random actions)
                const bool mid_ok = FMA(f_abs(R[8]), hc[2], FMA(f_abs(R[7]), hc[1], FMA(f_abs(R[6]), hc[0], cp[2]))) > m.middle_check_z;
                if (__builtin_amdgcn_ballot_w64(mid_ok) != 0ull) {
                    float a[3], b[3];
                    to_cube(L2{}, m.shape2.a, a);
                    to_cube(L2{}, m.shape2.b, b);
                    try_shape(L2{}, m.shape2, a, b, mid_ok);
                }
                try_sphere(L2{}, m.sph2[1], true);              // (its joint-3 housing comes down to 0.096 m: always looked at)
                // the joint-2 housing of the middle link and the upper link hang at the height of the base (0.29 m): only a cube above
                // upper_check_z can reach them
                const bool upper_ok = cp[2] > m.upper_check_z;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(upper_ok) != 0ull, 0)) {      // wave-level: practically never
                    try_sphere(L2{}, m.sph2[0], upper_ok);
                    float a[3], b[3];
                    to_cube(L1{}, m.shape1.a, a);
                    to_cube(L1{}, m.shape1.b, b);
                    try_shape(L1{}, m.shape1, a, b, upper_ok);
                }
            }
            // ---- FF on the finger wavefronts (cube kernels; TfModel.ff_middle_pairs): the MIDDLE link of this finger against the distal capsule of
            // each other finger.  Since API 8 these six rows are solved on the FREE velocities (oracle/tf_oracle.c: a Jacobi step), so a row needs only
            // what the fingers published before S1: this wavefront builds the two rows of its own middle link - out of its own frames, where the cube
            // role had to rebuild the middle frame from what is published - while the cube role runs the distal pairs, keeps its own velocity change
            // and posts the other finger's (ffm_mbox).  The same lines as the cube role's placement, the same bits. ----
            float ffm_own[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
            // Placement is a choice per instantiation (same bits either way): the 256-register cube kernels build the rows HERE (8192 envs: 46.9 -> 43.1 us;
            // one wavefront per SIMD, the cube wavefront is the long pole before the sweeps); the 128-register kernels keep them on the cube wavefront
            // (65536 envs: 70.4 us there against 73.0 us here - this block costs the finger role registers it has to spill), and so do the box kernels.
            // The 128-register kernels build only the SECOND group here (fd = f + 1: one row per finger wavefront) and leave the first to the cube role:
            // all six on the fingers cost this role registers it has to spill (65536 envs: 73.0 us against 70.0 us with all six on the cube wavefront).
            const bool ffm_here = (!BOXK || WIDE) && m.ff_middle_pairs != 0;     // wave-uniform
            constexpr int FFM_O_FIRST = WIDE ? 2 : 1;                  // this wavefront's groups: o = FFM_O_FIRST .. 1
            if (ffm_here && !HELP) {                                   // (HELP: wavefront 4 + f builds them, helper_role)
                const TfLinkShape& sh = m.shape2;
                const float jx = m.j3_origin[0], jy = m.j3_origin[1], jz = m.j3_origin[2];
                const float inv_j = f_rcp(FMA(jy, jy, jz * jz));
                const float ex[3] = {k.c1, 0.0f, -k.s1};
                float ey[3], aw[3], bw[3];
                {
                    float g[3], xg[3], ez[3], ab[3], bb[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) g[i] = FMA(-jx, ex[i], k.p3[i] - k.p2[i]);
                    cross3(ex, g, xg);
#pragma unroll
                    for (int i = 0; i < 3; ++i) { ey[i] = FMA(jy, g[i], -(jz * xg[i])) * inv_j; ez[i] = FMA(jz, g[i], jy * xg[i]) * inv_j; }
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        ab[i] = FMA(sh.a[2], ez[i], FMA(sh.a[1], ey[i], FMA(sh.a[0], ex[i], k.p2[i])));
                        bb[i] = FMA(sh.b[2], ez[i], FMA(sh.b[1], ey[i], FMA(sh.b[0], ex[i], k.p2[i])));
                    }
                    base_to_world(yw, ab, aw);
                    base_to_world(yw, bb, bw);
                }
                const float rest_ff = m.restitution_ff * dr[5];
#pragma unroll 1
                for (int o = FFM_O_FIRST; o >= 1; --o) {
                    const int fd = (f + o >= 3) ? f + o - 3 : f + o;
                    float dd[3] = {0.0f, 0.0f, 0.0f};
                    float Pm[3], Pd[3], sp;
                    {
                        float Ad[3], Bd[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) { Ad[j] = LD(L_REC(fd) + P_AW + j); Bd[j] = LD(L_REC(fd) + P_BW + j); }
                        seg_seg_s(aw, bw, Ad, Bd, Pm, Pd, sp);
                    }
                    const float dv[3] = {Pd[0] - Pm[0], Pd[1] - Pm[1], Pd[2] - Pm[2]};
                    const float dist2 = dot3(dv, dv);
                    const float inv = f_rsqrt(f_max(dist2, 1e-12f));
                    const float dist = dist2 * inv;
                    const float n[3] = {dv[0] * inv, dv[1] * inv, dv[2] * inv};      // from the middle link to the distal capsule
                    float nm[3];
                    dir_world_to_base(yw, n, nm);
                    const float u1 = dot3(nm, ex), u2 = dot3(nm, ey);
                    const float rho = FMA(sp, sh.rho[1] - sh.rho[0], sh.rho[0]);
                    const float h1 = FMA(sp, sh.w1[1] - sh.w1[0], sh.w1[0]) - rho, h2 = FMA(sp, sh.w2[1] - sh.w2[0], sh.w2[0]) - rho;
                    const float o1 = FMA(sp, sh.o1[1] - sh.o1[0], sh.o1[0]), o2 = FMA(sp, sh.o2[1] - sh.o2[0], sh.o2[0]);
                    const float ext = FMA(o2, u2, FMA(o1, u1, FMA(h2, f_abs(u2), FMA(h1, f_abs(u1), rho))));
                    const float gap_ff = dist - ext - m.cap_radius;
                    const bool near_ff = (dist2 > 1e-12f) && (gap_ff < m.contact_margin);
                    if (__builtin_amdgcn_ballot_w64(near_ff) != 0ull) {
                        float Jd[3], Wd[3], Jm[3], Wm[3], vd[3];
                        {   // distal side: the point of the capsule surface that faces the middle link
                            const Yaw yd = {m.base_yaw_cos[fd], m.base_yaw_sin[fd], 0.0f, 0.0f, m.base_height};
                            FingerPubRegs pd;
                            read_pub(lds, lane, fd, pd);
                            float C[3], Cb_[3], L1[3], L2[3], L3[3], nb[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) C[j] = FMA(-m.cap_radius, n[j], Pd[j]);
                            world_to_base(yd, C, Cb_);
                            levers(pd.k, Cb_, L1, L2, L3);
                            dir_world_to_base(yd, n, nb);
                            Jd[0] = dot3(L1, nb); Jd[1] = dot3(L2, nb); Jd[2] = dot3(L3, nb);
                            sym3_mul(pd.k.Minv, Jd, Wd);
                        }
                        {   // middle side: joints 1 and 2 move it, joint 3 does not
                            float C[3], Cb_[3], L1[3], L2[3], L3[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) C[j] = FMA(ext, n[j], Pm[j]);
                            world_to_base(yw, C, Cb_);
                            levers(k, Cb_, L1, L2, L3);
                            Jm[0] = dot3(L1, nm); Jm[1] = dot3(L2, nm); Jm[2] = 0.0f;
                            sym3_mul(k.Minv, Jm, Wm);
                        }
#pragma unroll
                        for (int j = 0; j < 3; ++j) vd[j] = LD(L_REC(fd) + P_VQ + j);          // free velocities on both sides
                        const float vn0 = dot3(Jd, vd) - dot3(Jm, vq);
                        if (near_ff && contact_live(m, gap_ff, vn0, h)) {
                            const float bias = contact_bias(m, gap_ff, vn0, inv_h, rest_ff);
                            const float lam = f_max(-(vn0 + bias) * f_rcp2(dot3(Jd, Wd) + dot3(Jm, Wm)), 0.0f);
#pragma unroll
                            for (int j = 0; j < 3; ++j) { dd[j] = Wd[j] * lam; ffm_own[o - 1][j] = Wm[j] * lam; }
                        }
                    }
                    // fd's mailbox: message 0 is the row of the first group (o = 2), message 1 the row of the second (o = 1)
                    const int msg = 2 - o;
                    if (fd == 0) { for (int j = 0; j < 3; ++j) LD(ffm_mbox(0, 3 * msg + j, BOXK && WIDE)) = dd[j]; }
                    else if (fd == 1) { for (int j = 0; j < 3; ++j) LD(ffm_mbox(1, 3 * msg + j, BOXK && WIDE)) = dd[j]; }
                    else { for (int j = 0; j < 3; ++j) LD(ffm_mbox(2, 3 * msg + j, BOXK && WIDE)) = dd[j]; }
                }
            }
            STAMP(sb_ + 2);
            BAR();                                              // S1b: the finger-finger pass of the cube role is done
            STAMP(sb_ + 3);
            // velocity after the finger-finger pass (the approach speeds below use the free velocity vq, as specified)
            float vqf[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) vqf[j] = LD(L_VQFF + 3 * f + j);
            if (ffm_here) {
                // the velocity changes of the rows in the order they are listed - first group (0;2) (1;0) (2;1), second group (0;1) (1;2) (2;0) - a zero
                // component skipped.  Within a group the rows are ordered by the finger that owns the middle link: this finger's own row (fm = f) and the
                // one it received (first group: fm = f + 1, second group: fm = f + 2, mod 3).  (128-register kernels: the first group is in L_VQFF already.)
                float mail[2][3];
                if (HELP) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) { ffm_own[0][j] = LD(ffm_own_mbox(f, j)); ffm_own[1][j] = LD(ffm_own_mbox(f, 3 + j)); }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    mail[0][j] = !WIDE ? 0.0f : ((f == 0) ? LD(ffm_mbox(0, j, BOXK && WIDE)) : ((f == 1) ? LD(ffm_mbox(1, j, BOXK && WIDE)) : LD(ffm_mbox(2, j, BOXK && WIDE))));
                    mail[1][j] = (f == 0) ? LD(ffm_mbox(0, 3 + j, BOXK && WIDE)) : ((f == 1) ? LD(ffm_mbox(1, 3 + j, BOXK && WIDE)) : LD(ffm_mbox(2, 3 + j, BOXK && WIDE)));
                }
                auto take = [&](const float d[3], bool minus) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) { const float t = minus ? vqf[j] - d[j] : vqf[j] + d[j]; vqf[j] = (d[j] != 0.0f) ? t : vqf[j]; }
                };
                if (WIDE) {                                          // first group: own row (f; f + 2), received row (f + 1; f)
                    if (f == 2) { take(mail[0], false); take(ffm_own[1], true); } else { take(ffm_own[1], true); take(mail[0], false); }
                }
                if (f == 0) { take(ffm_own[0], true); take(mail[1], false); }      // second group: own row (f; f + 1), received row (f + 2; f)
                else { take(mail[1], false); take(ffm_own[0], true); }
            }
            // rows of the finger-cube contact; the contact-space record goes straight to LDS for the cube role
            const int rb = L_REC(f);
            float fcJ[9], fc_arm[3], rec_lam[3];
            int cur_link = 0;
#pragma unroll
            for (int j = 0; j < 9; ++j) fcJ[j] = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) { fc_arm[j] = 0.0f; rec_lam[j] = 0.0f; }
            bool fc_live = false;
            if (__builtin_expect(gap < m.contact_margin, 1)) {
                float rcv[3], xw[3], Dd[3], J[9], W[9], dir[9], rxd[9];
                mat3_mul(R, nc, &dir[0]);
                mat3_mul(R, y, rcv);
                mat3_mul(R, x, xw);
                tangent_basis(&dir[0], &dir[3], &dir[6]);
                float Pw[3] = {FMA(-radius, dir[0], cp[0] + xw[0]), FMA(-radius, dir[1], cp[1] + xw[1]), FMA(-radius, dir[2], cp[2] + xw[2])};
                float Pb[3];
                world_to_base(yw, Pw, Pb);
                finger_jac(yw, k, link, Pb, dir, J, W, Dd);
#pragma unroll
                for (int d = 0; d < 3; ++d) cross3(rcv, &dir[3 * d], &rxd[3 * d]);
                if (__builtin_expect(box, 0)) {                 // general box: inertia-scaled arms S (r x d)
                    float S[6];
#pragma unroll
                    for (int e = 0; e < 6; ++e) S[e] = LD(L_POSE_S + e);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float t[3];
                        sym3_mul(S, &rxd[3 * d], t);
                        rxd[3 * d] = t[0]; rxd[3 * d + 1] = t[1]; rxd[3 * d + 2] = t[2];
                    }
                }
                float vn0 = dot3(&J[0], vq) - (dot3(&dir[0], cvf) + dot3(&rxd[0], cwf));
                if (contact_live(m, gap, vn0, h)) {
                    cur_link = link;
                    fc_live = true;
#pragma unroll
                    for (int j = 0; j < 9; ++j) { fcJ[j] = J[j]; LD(rb + R_DIR + j) = dir[j]; LD(rb + R_RXD + j) = rxd[j]; }
#pragma unroll
                    for (int d = 0; d < 3; ++d) LD(L_VQFF + 3 * f + d) = f_rcp2(FMA(dot3(&rxd[3 * d], &rxd[3 * d]), inv_I, Dd[d] + inv_m));
                    LD(rb + R_A + 0) = FMA(dot3(&rxd[0], &rxd[3]), inv_I, dot3(&J[0], &W[3]));
                    LD(rb + R_A + 1) = FMA(dot3(&rxd[0], &rxd[6]), inv_I, dot3(&J[0], &W[6]));
                    LD(rb + R_A + 2) = FMA(dot3(&rxd[3], &rxd[6]), inv_I, dot3(&J[3], &W[6]));
                    if (link == 3) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) fc_arm[j] = Pw[j] - Tw[j];
                    }
                    LD(L_INIT + f) = contact_bias(m, gap, vn0, inv_h, rest_f);
                    if ((float)link == fc_link) {                   // same link as in the last substep: seed the impulses
                        float l0 = lam_fc[0] * ws;
                        float lim = mu_fc * l0;
                        rec_lam[0] = l0;
                        rec_lam[1] = f_clamp(dot3(&lam_fc[1], &dir[3]) * ws, -lim, lim);
                        rec_lam[2] = f_clamp(dot3(&lam_fc[1], &dir[6]) * ws, -lim, lim);
                    }
                }
            }
            if (!fc_live) {                                     // a dead slot: 1/D = 0 tells the cube role
#pragma unroll
                for (int d = 0; d < 3; ++d) LD(L_VQFF + 3 * f + d) = 0.0f;
                LD(L_INIT + f) = 0.0f;
            }
            // fingertip sphere vs floor (slot 0) and vs boundary wall (slot 1)
            TipContact tc[2];
            float wall_n[3];                                    // inward surface normal of the boundary at the fingertip (rows and wrench of slot 1)
            {
                // fingertip sphere centre in the world (z) and relative to the stage centre (x, y)
                const float bx = EXT ? (Bw[0] + boff[0]) - soff[0] : Bw[0], by = EXT ? (Bw[1] + boff[1]) - soff[1] : Bw[1], bz = EXT ? Bw[2] + boff[2] : Bw[2];
                float rho2 = FMA(bx, bx, by * by);
                float inv = f_rsqrt(f_max(rho2, 1e-24f));
                float rho = rho2 * inv;
                float wc, wsn;
                const float wgap = (wall_profile(P, bz, wc, wsn) - rho) * wc;      // distance of the sphere centre to the (tilted) surface
                wall_n[0] = (-bx * inv) * wc; wall_n[1] = (-by * inv) * wc; wall_n[2] = wsn;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    TipContact& c = tc[t];
                    c.active = false;
#pragma unroll
                    for (int j = 0; j < 9; ++j) c.J[j] = 0.0f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) { c.Dinv[j] = 0.0f; c.lam[j] = 0.0f; c.arm[j] = 0.0f; }
                    c.bias = 0.0f; c.mu = 0.0f;
                    float gp_ = (t == 0) ? (bz - m.cap_radius) : (wgap - m.cap_radius);
                    const bool on = ((t == 0) || (rho > 1e-6f)) && (gp_ < m.contact_margin);
                    if (__builtin_expect(on, t == 0)) {
                        float dir[9] = {0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                        if (t == 1) { dir[0] = wall_n[0]; dir[1] = wall_n[1]; dir[2] = wall_n[2]; }
                        tangent_basis(&dir[0], &dir[3], &dir[6]);
                        float Pw[3] = {FMA(-m.cap_radius, dir[0], Bw[0]), FMA(-m.cap_radius, dir[1], Bw[1]), FMA(-m.cap_radius, dir[2], Bw[2])};
                        float Pb[3], Dd[3], Jt[9], Wt[9];
                        world_to_base(yw, Pw, Pb);
                        finger_jac(yw, k, 3, Pb, dir, Jt, Wt, Dd);
                        float vn0 = dot3(&Jt[0], vq);
                        if (contact_live(m, gp_, vn0, h)) {
                            c.active = true;
                            c.mu = (t == 0) ? mu_tf : mu_tw;
#pragma unroll
                            for (int j = 0; j < 9; ++j) c.J[j] = Jt[j];
#pragma unroll
                            for (int d = 0; d < 3; ++d) c.Dinv[d] = f_rcp2(Dd[d]);
#pragma unroll
                            for (int j = 0; j < 3; ++j) c.arm[j] = Pw[j] - Tw[j];
                            c.bias = contact_bias(m, gp_, vn0, inv_h, rest_f);
                            const float* pl = (t == 0) ? lam_tf : lam_tw;      // zero when the contact was not there
                            float l0 = pl[0] * ws;
                            float lim = c.mu * l0;
                            c.lam[0] = l0;
                            c.lam[1] = f_clamp(pl[1] * ws, -lim, lim);
                            c.lam[2] = f_clamp(pl[2] * ws, -lim, lim);
                        }
                    }
                }
            }
            // joint limit / velocity limit rows
            float vlo[3], vhi[3], lim_dinv[3], lim_lam[3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int dg = (jj == 0) ? 0 : ((jj == 1) ? 3 : 5);
                vlo[jj] = f_clamp((m.q_lo[jj] - q[jj]) * inv_h, -m.qd_max, m.qd_max);
                vhi[jj] = f_clamp((m.q_hi[jj] - q[jj]) * inv_h, -m.qd_max, m.qd_max);
                lim_dinv[jj] = f_rcp(k.Minv[dg]);
                lim_lam[jj] = 0.0f;
            }
            // ---- seeded impulses on the finger side, contact-point velocity ----
#pragma unroll
            for (int j = 0; j < 3; ++j) vq[j] = vqf[j];
            if (cur_link != 0) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    float Wd[3];
                    sym3_mul(k.Minv, &fcJ[3 * d], Wd);
#pragma unroll
                    for (int j = 0; j < 3; ++j) vq[j] = FMA(Wd[j], rec_lam[d], vq[j]);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (tc[t].active) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float Wd[3];
                        sym3_mul(k.Minv, &tc[t].J[3 * d], Wd);
#pragma unroll
                        for (int j = 0; j < 3; ++j) vq[j] = FMA(Wd[j], tc[t].lam[d], vq[j]);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                LD(rb + R_U + d) = dot3(&fcJ[3 * d], vq);
                LD(rb + R_DL + d) = rec_lam[d];                 // seeded impulses (the cube role reads them once)
            }
            STAMP(sb_ + 4);
            BAR();                                              // S3: records published
            STAMP(sb_ + 5);
            // ---- projected Gauss-Seidel: this finger's share ----
            float Fc[3] = {0.0f, 0.0f, 0.0f};
            uint32_t t_wait = 0u;
            for (int it = 0; it < P.iters; ++it) {
                { const uint32_t t0_ = NOW(); BAR(); t_wait += NOW() - t0_; }   // W1: the cube role has solved the finger-cube rows of this sweep
                float dl[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) dl[d] = LD(rb + R_DL + d);
                if (cur_link != 0) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float Wd[3];
                        sym3_mul(k.Minv, &fcJ[3 * d], Wd);
#pragma unroll
                        for (int j = 0; j < 3; ++j) vq[j] = FMA(Wd[j], dl[d], vq[j]);
                    }
                }
                // TfConfig.solver_inner > 1: the finger-only rows get their turn on every inner-th pass only (the passes in between visit the block of all
                // rows that touch the cube; this finger follows them through its contact-point velocity)
                const bool own_rows = (P.inner == 1) || (((it + 1) % P.inner) == 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {                   // fingertip - floor, fingertip - wall
                    TipContact& c = tc[t];
                    if (c.active && own_rows) {
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            float vrel = dot3(&c.J[3 * d], vq);
                            float dlt = (d == 0) ? solve_normal(c.lam[0], c.Dinv[0], vrel, c.bias)
                                                 : solve_tangent(c.lam[d], c.Dinv[d], vrel, c.mu * c.lam[0]);
                            float Wd[3];
                            sym3_mul(k.Minv, &c.J[3 * d], Wd);
#pragma unroll
                            for (int j = 0; j < 3; ++j) vq[j] = FMA(Wd[j], dlt, vq[j]);
                        }
                    }
                }
                if (own_rows) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {                // joint limits + velocity limit
                    const int dg = (jj == 0) ? 0 : ((jj == 1) ? 3 : 5);
                    const int c0 = (jj == 0) ? 0 : ((jj == 1) ? 1 : 2);
                    const int c1 = (jj == 0) ? 1 : ((jj == 1) ? 3 : 4);
                    const int c2 = (jj == 0) ? 2 : ((jj == 1) ? 4 : 5);
                    float v0 = FMA(-k.Minv[dg], lim_lam[jj], vq[jj]);
                    float tgt = f_clamp(v0, vlo[jj], vhi[jj]);
                    float lam_new = (tgt - v0) * lim_dinv[jj];
                    float dlj = lam_new - lim_lam[jj];
                    lim_lam[jj] = lam_new;
                    vq[0] = FMA(k.Minv[c0], dlj, vq[0]);
                    vq[1] = FMA(k.Minv[c1], dlj, vq[1]);
                    vq[2] = FMA(k.Minv[c2], dlj, vq[2]);
                }
                }
                if (cur_link != 0) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) LD(rb + R_U + d) = dot3(&fcJ[3 * d], vq);
                }
                { const uint32_t t0_ = NOW(); BAR(); t_wait += NOW() - t0_; }   // W2: contact-point velocities published
            }
            // impulses of the finger-cube contact after the last sweep (the cube role left them behind barrier W2 of that sweep; nobody touches
            // these slots again before this finger's own next publication)
            lam_fc[0] = LD(L_INIT + f);
#pragma unroll
            for (int j = 0; j < 3; ++j) { lam_fc[1 + j] = LD(rb + R_A + j); Fc[j] = LD(rb + R_A + 3 + j); }
            STAMP(sb_ + 6);
            STAMPV(sb_ + 8, t_wait);
            // ---- impulses kept for the next substep (state rows), fingertip wrench sensor, integration ----
            const bool last_sub = s == nsub - 1;                // wave-uniform: state rows after the last substep, LDS parking otherwise
            if (last_sub) {
                const bool tw_now = tc[1].lam[0] > 0.0f;
                STST(TF_S_FC_LINK + f, (float)(cur_link + (tw_now ? 4 : 0)));
                if (cur_link != 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) STST(TF_S_LAM_FC + 4 * f + j, lam_fc[j]);
                }
#pragma unroll
                for (int d = 0; d < 3; ++d) STST(TF_S_LAM_TF + 3 * f + d, tc[0].lam[d]);
                if (__builtin_expect(tw_now, 0)) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) STST(TF_S_LAM_TW + 3 * f + d, tc[1].lam[d]);
                }
            } else if (WIDE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) k_lam_fc[j] = lam_fc[j];
                k_fc_link = (float)cur_link;
#pragma unroll
                for (int d = 0; d < 3; ++d) { k_lam_tf[d] = tc[0].lam[d]; k_lam_tw[d] = tc[1].lam[d]; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) LD(L_PARK(f) + PK_FC + j) = lam_fc[j];
                LD(L_PARK(f) + PK_LINK) = (float)cur_link;
#pragma unroll
                for (int d = 0; d < 3; ++d) { LD(L_PARK(f) + PK_TF + d) = tc[0].lam[d]; LD(L_PARK(f) + PK_TW + d) = tc[1].lam[d]; LD(L_PARK(f) + PK_TAU + d) = tau[d]; }
            }
            if (ASYM) {
                float* ft = ft_run;
                if (cur_link == 3) {
                    float T[3];
                    cross3(fc_arm, Fc, T);
#pragma unroll
                    for (int j = 0; j < 3; ++j) { ft[j] += Fc[j]; ft[3 + j] += T[j]; }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const TipContact& c = tc[t];
                    if (c.active) {
                        float dir[9] = {0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};      // same directions the rows were built with
                        if (t == 1) { dir[0] = wall_n[0]; dir[1] = wall_n[1]; dir[2] = wall_n[2]; }
                        tangent_basis(&dir[0], &dir[3], &dir[6]);
                        float F[3], T[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) F[j] = FMA(dir[6 + j], c.lam[2], FMA(dir[3 + j], c.lam[1], dir[j] * c.lam[0])) * inv_h;
                        cross3(c.arm, F, T);
#pragma unroll
                        for (int j = 0; j < 3; ++j) { ft[j] += F[j]; ft[3 + j] += T[j]; }
                    }
                }
                if (last_sub) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) { if (FT_IN_REGS) { if (WIDE) k_ft[j] = ft[j]; else LD(L_PARK(f) + PK_FT + j) = ft[j]; } else STST(TF_S_FT + 6 * f + j, ft[j]); }
                } else {
#pragma unroll
                    for (int j = 0; j < 6; ++j) { if (WIDE) k_ft[j] = ft[j]; else LD(L_PARK(f) + PK_FT + j) = ft[j]; }
                }
            }
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                qd[jj] = vq[jj];
                q[jj] = f_clamp(FMA(h, vq[jj], q[jj]), m.q_lo[jj], m.q_hi[jj]);
            }
            STAMP(sb_ + 7);
        }
    }
    STAMP(30);
    // =================================================================================================================
    // post: fingertip state, observation slots of this finger
    // =================================================================================================================
    bool f_guarded = false;                                    // NaN guard of the env (set in the post phase; read by the presampler below)
    if (MODE & M_POST) {
        FK pk;
        float tips[13];
        fk_setup(m, q, pk);
        tip_state(m, yw, pk, q, qd, tips);
        float boff_p[3];                                       // robot base offset of the episode (cold through the physics: re-read)
#pragma unroll
        for (int j = 0; j < 3; ++j) { boff_p[j] = (EXT && P.dr_enable) ? LDST(TF_S_DR + TF_DR_BASE_POS + j) : 0.0f; if (EXT) tips[j] = tips[j] + boff_p[j]; }   // robot frame -> world
        float tip_prev[3], tau_p[3], act_p[AJ], ft[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) { tip_prev[j] = LDST(TF_S_TIP_P + 3 * f + j); tau_p[j] = LDST(TF_S_TAU + 3 * f + j); }
        if (ASYM) {
#pragma unroll
            for (int j = 0; j < 6; ++j) ft[j] = FT_IN_REGS ? (WIDE ? k_ft[j] : LD(L_PARK(f) + PK_FT + j)) : LDST(TF_S_FT + 6 * f + j);   // (own slots, untouched until the tile is written behind P1)
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {         // the last command, as _action_buf holds it (written above by this workgroup)
            const int col = (j < 3) ? (3 * f + j) : (9 + 3 * f + (j - 3));
            act_p[j] = P.action_buf[(size_t)cx.i * (size_t)A + (size_t)col];
        }
        {   // NaN guard, part 1: this role's share of the finiteness test
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = acc + q[j] * 0.0f + qd[j] * 0.0f;
            LD(L_NAN + f) = (acc == 0.0f) ? 0.0f : 1.0f;
        }
        STAMP(31);
        BAR();                                                  // P1
        STAMP(32);
        const bool guarded = (LD(L_NAN) + LD(L_NAN + 1) + LD(L_NAN + 2) + LD(L_NAN + 3)) != 0.0f;
        f_guarded = guarded;
        if (__builtin_expect(guarded, 0)) {                     // park the env at the default pose (it is flagged for reset)
#pragma unroll
            for (int j = 0; j < 3; ++j) { q[j] = m.q_default[j]; qd[j] = 0.0f; }
#pragma unroll
            for (int j = 0; j < 6; ++j) { ft[j] = 0.0f; if (!FT_IN_REGS) STST(TF_S_FT + 6 * f + j, 0.0f); }
            fk_setup(m, q, pk);
            tip_state(m, yw, pk, q, qd, tips);
#pragma unroll
            for (int j = 0; j < 3; ++j) { if (EXT) tips[j] = tips[j] + boff_p[j]; }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) STST(TF_S_TIP_P + 3 * f + j, tips[j]);
        // exchange for the rewards (cube role) and this finger's slots of the obs / states tile
        const float co = P.clip_obs;
        const bool nrm = P.normalize_obs != 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) { LD(L_XCH + 6 * f + j) = tips[j]; LD(L_XCH + 6 * f + 3 + j) = tip_prev[j]; }
        if (!IS_RESET) {                                        // this finger's terms of the finger_move_penalty sum (rewards.py:165-184): the cube role adds them up in order
#pragma unroll
            for (int j = 0; j < 3; ++j) { const float vel = (tips[j] - tip_prev[j]) / sa.rc.dt; LD(L_VSQ + 3 * f + j) = vel * vel; }
        }
        float* row = &lds[lane * TW];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float lo = (j == 0) ? -0.33f : ((j == 1) ? 0.0f : -2.7f), hi = (j == 0) ? 1.0f : ((j == 1) ? 1.57f : 0.0f);
            row[3 * f + j] = emit_scaled(q[j], lo, hi, nrm, co);
            row[9 + 3 * f + j] = emit_scaled(qd[j], -10.0f, 10.0f, nrm, co);
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int col = (j < 3) ? (3 * f + j) : (9 + 3 * f + (j - 3));
            row[32 + col] = emit_table(P, 32 + col, opaque(act_p[j]), nrm, co);
        }
        if (ASYM) {
#pragma unroll
            for (int j = 0; j < 13; ++j) row[OD + 6 + 13 * f + j] = emit_scaled(tips[j], TLO(j), THI(j), nrm, co);
#pragma unroll
            for (int j = 0; j < 3; ++j) row[OD + 45 + 3 * f + j] = emit_scaled(P.enable_ft ? tau_p[j] : 0.0f, -0.36f, 0.36f, nrm, co);
            const float inv_n = 1.0f / (float)(P.substeps * P.control_decimation);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float wv[3] = {ft[3 * half] * inv_n, ft[3 * half + 1] * inv_n, ft[3 * half + 2] * inv_n};
                float bv[3], lv[3];
                dir_world_to_base(yw, wv, bv);
                rot_link_T<3>(pk, bv, lv);
#pragma unroll
                for (int j = 0; j < 3; ++j) row[OD + 54 + 6 * f + 3 * half + j] = emit_scaled(P.enable_ft ? lv[j] : 0.0f, -1.0f, 1.0f, nrm, co);
            }
        }
        if (ASYM) {
            STAMP(33);
            BAR();                                              // P3: states tile complete
            STAMP(34);
            coop_store_tile<SD, NT_F>(P.states, lds, cx);
            if (P.dr_obs_noise > 0.0f) {
                BAR();                                          // P4: states tile stored; obs noise goes on top of slots 0..24
                float nz[28];
#pragma unroll
                for (int b = 0; b < 28; ++b) nz[b] = 0.0f;
#pragma unroll
                for (int b = 0; b < 5; ++b) {                   // only the Philox blocks that hold this finger's six slots (wave-uniform: f is)
                    const int lo = 4 * b, hi = 4 * b + 3;
                    const bool need = (lo <= 3 * f + 2 && hi >= 3 * f) || (lo <= 11 + 3 * f && hi >= 9 + 3 * f);
                    if (need) rng4(P, gid, sa.frame, RNG_OBS_NOISE + (uint32_t)b, &nz[4 * b]);
                }
#pragma unroll
                for (int jj = 0; jj < 18; ++jj) {
                    const bool mine = (jj == 3 * f) || (jj == 3 * f + 1) || (jj == 3 * f + 2) || (jj == 9 + 3 * f) || (jj == 10 + 3 * f) || (jj == 11 + 3 * f);
                    if (mine) row[jj] = f_clamp(FMA(P.dr_obs_noise, 2.0f * nz[jj] - 1.0f, row[jj]), -co, co);
                }
                BAR();                                          // P5
            }
            coop_store_tile_strided<OD, SD, NT_F>(P.obs, lds, cx);
        } else {
            if (P.dr_obs_noise > 0.0f) {
                float nz[28];
#pragma unroll
                for (int b = 0; b < 28; ++b) nz[b] = 0.0f;
#pragma unroll
                for (int b = 0; b < 5; ++b) {                   // only the Philox blocks that hold this finger's six slots (wave-uniform: f is)
                    const int lo = 4 * b, hi = 4 * b + 3;
                    const bool need = (lo <= 3 * f + 2 && hi >= 3 * f) || (lo <= 11 + 3 * f && hi >= 9 + 3 * f);
                    if (need) rng4(P, gid, sa.frame, RNG_OBS_NOISE + (uint32_t)b, &nz[4 * b]);
                }
#pragma unroll
                for (int jj = 0; jj < 18; ++jj) {
                    const bool mine = (jj == 3 * f) || (jj == 3 * f + 1) || (jj == 3 * f + 2) || (jj == 9 + 3 * f) || (jj == 10 + 3 * f) || (jj == 11 + 3 * f);
                    if (mine) row[jj] = f_clamp(FMA(P.dr_obs_noise, 2.0f * nz[jj] - 1.0f, row[jj]), -co, co);
                }
            }
            BAR();                                              // P3
            coop_store_tile<OD, NT_F>(P.obs, lds, cx);
        }
    }
    // ---- state rows of this finger ----
    if (MODE & (M_RESETS | M_SIM | M_POST)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { STST(TF_S_Q + 3 * f + j, q[j]); STST(TF_S_QD + 3 * f + j, qd[j]); }
    }
    // ---- the samples of the NEXT reset of an env this step flags (time-out, non-finite state): drawn HERE, where the finger wavefronts have finished and
    // the cube wavefront still evaluates rewards and statistics - one third each: finger 0 the object pose and the tag, finger 1 the goal position,
    // finger 2 the goal orientation and angular velocity (include/trifinger.h: TF_S_NEXT_*; the cube role's reset block loads them) ----
    if (PRESAMPLE) {
        const int steps_now = (rflag ? 0 : fl_steps) + 1;                                        // what the cube role writes to steps[] at the end of this step
        const bool will_reset = f_guarded || (P.episode_length > 0 && steps_now >= P.episode_length);    // = reset_buf after this step
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(will_reset) != 0ull, 0)) {
            const uint32_t count = fl_count + (rflag ? 1u : 0u) + ((fl_goal_reset != 0) ? 1u : 0u);         // reset_count after this step's resets
            if (f == 0) {
                if (P.object_reset_type == TF_RESET_RANDOM) {
                    float u[4], ox, oy, oq[4];
                    rng4(P, gid, count, RNG_OBJECT, u);
                    sample_xy(u[0], u[1], m.obj_max_com_dist, ox, oy);
                    sample_yaw_quat(u[2], oq);
                    if (will_reset) { STST(TF_S_NEXT_OBJ + 0, ox); STST(TF_S_NEXT_OBJ + 1, oy); STST(TF_S_NEXT_OBJ + 2, oq[2]); STST(TF_S_NEXT_OBJ + 3, oq[3]); }
                }
                if (will_reset) STST(TF_S_NEXT_TAG, __builtin_bit_cast(float, count + 1u));
            } else if (f == 1) {
                float xyz[3], yq[4];
                sample_goal_pos(P, gid, count, xyz, yq);
                if (will_reset) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) STST(TF_S_NEXT_GOAL + j, xyz[j]);
                    if (!goal_has_random_quat(P)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) STST(TF_S_NEXT_GOAL + 3 + j, yq[j]);
                    }
                }
            } else {
                float rq[4] = {0.0f, 0.0f, 0.0f, 1.0f}, gwn[3];
                sample_goal_rot(P, gid, count, rq, gwn);
                if (will_reset) {
                    if (goal_has_random_quat(P)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) STST(TF_S_NEXT_GOAL + 3 + j, rq[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) STST(TF_S_NEXT_GOAL + 7 + j, gwn[j]);
                }
            }
        }
    }
    STAMP(35);
    STAMPV(40, __builtin_amdgcn_s_getreg((31 << 11) | 4));      // HW_REG_HW_ID
    STAMPV(41, __builtin_amdgcn_s_getreg((31 << 11) | 20));     // HW_REG_XCC_ID
}

// =====================================================================================================================
// HELPER ROLE (HELP instantiation: wavefronts 4..7 of an eight-wavefront workgroup)
// =====================================================================================================================
// Up to 16384 envs a 256-register workgroup has a CU to itself and one wavefront on each SIMD: the second wavefront slot of a SIMD is empty, and the six
// middle-distal finger-finger rows - independent of everything else between S1 and S1b, functions of what the fingers published before S1 - sat on the
// finger wavefronts' critical path (contact generation).  Here wavefront 4 + fm builds the two rows of finger fm's middle link with the cube role's
// formulation of them (the middle frame rebuilt from what finger fm publishes: the lines of cube_role's block, the same bits) and posts the velocity
// changes; the finger roles add them behind S1b in the listed order exactly as they add the ones they computed themselves.  Apart from that the role
// only keeps the workgroup's barriers company - every BAR() of the other roles has its twin here, in the same order under the same conditions.
template <bool ASYM, int MODE, int X>
DEV void helper_role(const DevParams& P, const StepArgs& sa, float* lds, const Ctx& cx) {
    constexpr bool BOXK = X == 2;
    const TfModel& m = P.m;
    const int lane = cx.lane;
    const int fm = cx.role - 4;
    BAR();                                                      // #1
    if (!(MODE & (M_ACT_IN | M_RESETS))) BAR();                 // #1b
    if (MODE & (M_ACT_IN | M_RESETS)) { BAR(); BAR(); }         // #2a, #2b
    if (MODE & M_SIM) {
        const float h = P.hsub, inv_h = 1.0f / h;
        const int nsub = sa.nsim * P.substeps;
        const TfLinkShape& sh = m.shape2;
        const float jx = m.j3_origin[0], jy = m.j3_origin[1], jz = m.j3_origin[2];
        const float inv_j = f_rcp(FMA(jy, jy, jz * jz));
        const int fmi = fm < 3 ? fm : 0;
        const Yaw ym = {m.base_yaw_cos[fmi], m.base_yaw_sin[fmi], 0.0f, 0.0f, m.base_height};
        for (int s = 0; s < nsub; ++s) {
            BAR();                                              // S1
            if (fm == 3) {
                // wavefront 7: the distal pass - the arithmetic of tf_ff_distal.inc (which the cube role runs when there are no helpers), laid out for a
                // wavefront that has nothing else to do: what the fingers published is read once, the geometry of the three pairs (closest points,
                // Jacobians, M^-1 J^T - independent of each other) is built side by side, and only the three velocity updates, each of which sees the
                // one before it, run in turn, on registers; L_VQFF is written once at the end.  Same operations on the same operands: the same bits.
                const float rest_ff = m.restitution_ff * LD(L_HELP_DR);
                float Aw_[3][3], Bw_[3][3], vel[3][3];
                FingerPubRegs pp[3];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) { Aw_[f][j] = LD(L_REC(f) + P_AW + j); Bw_[f][j] = LD(L_REC(f) + P_BW + j); vel[f][j] = LD(L_REC(f) + P_VQ + j); }
                    read_pub(lds, lane, f, pp[f]);
                }
                float gapp[3], Ja[3][3], Wa[3][3], Jb[3][3], Wb[3][3];
                bool nearp[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const int fa = p, fb = (p == 2) ? 0 : p + 1;
                    float Pa[3], Pb[3];
                    seg_seg(Aw_[fa], Bw_[fa], Aw_[fb], Bw_[fb], Pa, Pb);
                    float dv[3] = {Pa[0] - Pb[0], Pa[1] - Pb[1], Pa[2] - Pb[2]};
                    float dist2 = dot3(dv, dv);
                    float inv = f_rsqrt(f_max(dist2, 1e-12f));
                    float dist = dist2 * inv;
                    gapp[p] = dist - 2.0f * m.cap_radius;
                    nearp[p] = (dist2 > 1e-12f) && (gapp[p] < m.contact_margin);
#pragma unroll
                    for (int j = 0; j < 3; ++j) { Ja[p][j] = 0.0f; Wa[p][j] = 0.0f; Jb[p][j] = 0.0f; Wb[p][j] = 0.0f; }
                    if (__builtin_amdgcn_ballot_w64(nearp[p]) != 0ull) {
                        float n[3] = {dv[0] * inv, dv[1] * inv, dv[2] * inv};       // from finger b to finger a
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            const int ff_ = side ? fb : fa;
                            const Yaw yy = {m.base_yaw_cos[ff_], m.base_yaw_sin[ff_], 0.0f, 0.0f, m.base_height};
                            float C[3], Cb_[3], L1[3], L2[3], L3[3], nb[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) C[j] = side ? FMA(m.cap_radius, n[j], Pb[j]) : FMA(-m.cap_radius, n[j], Pa[j]);
                            world_to_base(yy, C, Cb_);
                            levers(pp[ff_].k, Cb_, L1, L2, L3);
                            dir_world_to_base(yy, n, nb);
                            float* J = side ? Jb[p] : Ja[p];
                            float* W = side ? Wb[p] : Wa[p];
                            J[0] = dot3(L1, nb); J[1] = dot3(L2, nb); J[2] = dot3(L3, nb);
                            sym3_mul(pp[ff_].k.Minv, J, W);
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const int fa = p, fb = (p == 2) ? 0 : p + 1;
                    if (nearp[p]) {
                        const float vn0 = dot3(Ja[p], vel[fa]) - dot3(Jb[p], vel[fb]);
                        if (contact_live(m, gapp[p], vn0, h)) {
                            const float bias = contact_bias(m, gapp[p], vn0, inv_h, rest_ff);
                            const float lam = f_max(-(vn0 + bias) * f_rcp2(dot3(Ja[p], Wa[p]) + dot3(Jb[p], Wb[p])), 0.0f);
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                vel[fa][j] = FMA(Wa[p][j], lam, vel[fa][j]);
                                vel[fb][j] = FMA(-Wb[p][j], lam, vel[fb][j]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int f = 0; f < 3; ++f) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) LD(L_VQFF + 3 * f + j) = vel[f][j];
                }
            } else if (m.ff_middle_pairs != 0) {
                const float rest_ff = m.restitution_ff * LD(L_HELP_DR + fm);
                FingerPubRegs pm;
                read_pub(lds, lane, fm, pm);
                const float ex[3] = {pm.k.c1, 0.0f, -pm.k.s1};
                float ey[3], aw[3], bw[3], vm[3];
                {
                    float g[3], xg[3], ez[3], ab[3], bb[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) g[i] = FMA(-jx, ex[i], pm.k.p3[i] - pm.k.p2[i]);
                    cross3(ex, g, xg);
#pragma unroll
                    for (int i = 0; i < 3; ++i) { ey[i] = FMA(jy, g[i], -(jz * xg[i])) * inv_j; ez[i] = FMA(jz, g[i], jy * xg[i]) * inv_j; }
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        ab[i] = FMA(sh.a[2], ez[i], FMA(sh.a[1], ey[i], FMA(sh.a[0], ex[i], pm.k.p2[i])));
                        bb[i] = FMA(sh.b[2], ez[i], FMA(sh.b[1], ey[i], FMA(sh.b[0], ex[i], pm.k.p2[i])));
                    }
                    base_to_world(ym, ab, aw);
                    base_to_world(ym, bb, bw);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) vm[j] = LD(L_REC(fm) + P_VQ + j);      // the FREE velocities (API 8)
#pragma unroll 1
                for (int o = 2; o >= 1; --o) {
                    const int fd = (fm + o >= 3) ? fm + o - 3 : fm + o;
                    float dd[3] = {0.0f, 0.0f, 0.0f}, dm[3] = {0.0f, 0.0f, 0.0f};
                    float Pm[3], Pd[3], sp;
                    {
                        float Ad[3], Bd[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) { Ad[j] = LD(L_REC(fd) + P_AW + j); Bd[j] = LD(L_REC(fd) + P_BW + j); }
                        seg_seg_s(aw, bw, Ad, Bd, Pm, Pd, sp);
                    }
                    const float dv[3] = {Pd[0] - Pm[0], Pd[1] - Pm[1], Pd[2] - Pm[2]};
                    const float dist2 = dot3(dv, dv);
                    const float inv = f_rsqrt(f_max(dist2, 1e-12f));
                    const float dist = dist2 * inv;
                    const float n[3] = {dv[0] * inv, dv[1] * inv, dv[2] * inv};      // from the middle link to the distal capsule
                    float nm[3];
                    dir_world_to_base(ym, n, nm);
                    const float u1 = dot3(nm, ex), u2 = dot3(nm, ey);
                    const float rho = FMA(sp, sh.rho[1] - sh.rho[0], sh.rho[0]);
                    const float h1 = FMA(sp, sh.w1[1] - sh.w1[0], sh.w1[0]) - rho, h2 = FMA(sp, sh.w2[1] - sh.w2[0], sh.w2[0]) - rho;
                    const float o1 = FMA(sp, sh.o1[1] - sh.o1[0], sh.o1[0]), o2 = FMA(sp, sh.o2[1] - sh.o2[0], sh.o2[0]);
                    const float ext = FMA(o2, u2, FMA(o1, u1, FMA(h2, f_abs(u2), FMA(h1, f_abs(u1), rho))));
                    const float gap = dist - ext - m.cap_radius;
                    const bool near_ff = (dist2 > 1e-12f) && (gap < m.contact_margin);
                    if (__builtin_amdgcn_ballot_w64(near_ff) != 0ull) {
                        float Jd[3], Wd[3], Jm[3], Wm[3], vd[3];
                        {   // distal side: the point of the capsule surface that faces the middle link
                            const Yaw yd = {m.base_yaw_cos[fd], m.base_yaw_sin[fd], 0.0f, 0.0f, m.base_height};
                            FingerPubRegs pd;
                            read_pub(lds, lane, fd, pd);
                            float C[3], Cb_[3], L1[3], L2[3], L3[3], nb[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) C[j] = FMA(-m.cap_radius, n[j], Pd[j]);
                            world_to_base(yd, C, Cb_);
                            levers(pd.k, Cb_, L1, L2, L3);
                            dir_world_to_base(yd, n, nb);
                            Jd[0] = dot3(L1, nb); Jd[1] = dot3(L2, nb); Jd[2] = dot3(L3, nb);
                            sym3_mul(pd.k.Minv, Jd, Wd);
                        }
                        {   // middle side: joints 1 and 2 move it, joint 3 does not
                            float C[3], Cb_[3], L1[3], L2[3], L3[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) C[j] = FMA(ext, n[j], Pm[j]);
                            world_to_base(ym, C, Cb_);
                            levers(pm.k, Cb_, L1, L2, L3);
                            Jm[0] = dot3(L1, nm); Jm[1] = dot3(L2, nm); Jm[2] = 0.0f;
                            sym3_mul(pm.k.Minv, Jm, Wm);
                        }
#pragma unroll
                        for (int j = 0; j < 3; ++j) vd[j] = LD(L_REC(fd) + P_VQ + j);
                        const float vn0 = dot3(Jd, vd) - dot3(Jm, vm);
                        if (near_ff && contact_live(m, gap, vn0, h)) {
                            const float bias = contact_bias(m, gap, vn0, inv_h, rest_ff);
                            const float lam = f_max(-(vn0 + bias) * f_rcp2(dot3(Jd, Wd) + dot3(Jm, Wm)), 0.0f);
#pragma unroll
                            for (int j = 0; j < 3; ++j) { dd[j] = Wd[j] * lam; dm[j] = Wm[j] * lam; }
                        }
                    }
                    // fd's mailbox: message 0 is the row of the first group (o = 2), message 1 the row of the second (o = 1); fm's own: by group
                    const int msg = 2 - o;
                    if (fd == 0) { for (int j = 0; j < 3; ++j) LD(ffm_mbox(0, 3 * msg + j, BOXK)) = dd[j]; }
                    else if (fd == 1) { for (int j = 0; j < 3; ++j) LD(ffm_mbox(1, 3 * msg + j, BOXK)) = dd[j]; }
                    else { for (int j = 0; j < 3; ++j) LD(ffm_mbox(2, 3 * msg + j, BOXK)) = dd[j]; }
#pragma unroll
                    for (int j = 0; j < 3; ++j) LD(ffm_own_mbox(fm, 3 * (o - 1) + j)) = dm[j];
                }
            }
            BAR();                                              // S1b
            BAR();                                              // S3
            for (int it = 0; it < P.iters; ++it) { BAR(); BAR(); }     // W1, W2
        }
    }
    if (MODE & M_POST) {
        BAR();                                                  // P1
        BAR();                                                  // P3
        if (ASYM && P.dr_obs_noise > 0.0f) { BAR(); BAR(); }    // P4, P5
    }
}

// =====================================================================================================================
// CUBE ROLE: cube, goal, flags and counters, rewards, termination, episode statistics
// =====================================================================================================================
struct LaneStats { float rew[6]; float pos_cnt, ori_cnt, succ, resets, nonfinite; };


// Sum over the 64 lanes, valid in lane 63.  DPP only (operand swizzles of v_add_f32, no LDS round trips like
// ds_bpermute): xor-1 / xor-2 inside the quads, rotate by 4 and 8 inside the 16-lane rows, then row_bcast:15 into
// rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Fixed order, so the statistics stay deterministic.
template <int CTRL, int ROW_MASK>
DEV float dpp_add(float x) {
    int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, false);
    return x + __builtin_bit_cast(float, moved);
}
DEV float wave_sum_lane63(float x) {
    x = dpp_add<0xB1, 0xF>(x);      // quad_perm [1,0,3,2]
    x = dpp_add<0x4E, 0xF>(x);      // quad_perm [2,3,0,1]
    x = dpp_add<0x124, 0xF>(x);     // row_ror:4
    x = dpp_add<0x128, 0xF>(x);     // row_ror:8
    x = dpp_add<0x142, 0xA>(x);     // row_bcast:15 -> rows 1, 3
    x = dpp_add<0x143, 0xC>(x);     // row_bcast:31 -> rows 2, 3
    return x;
}


// Episode statistics without a second launch (a separate 11-wave reduction kernel cost 4.5 us per step, mostly fixed
// launch and cold-miss latency).  Lane k < 11 of every wave adds the wave's sum of statistic k to a 64-bit accumulator with
// ONE device-scope integer atomic: the sum as signed fixed point (2^-16) in the upper 47 bits, an arrival count in the
// lower 17.  Integer addition commutes, so the result does not depend on the arrival order (deterministic, unlike float
// atomics), and the returned old value tells each lane whether it was the last to arrive: that lane carries the total to
// the next level (16 shards -> 1, so that a thousand waves finishing together do not queue on one word) and finally
// writes info[].  The atomic is issued as soon as the rewards are known and its return is consumed at the very end of
// the kernel, behind the observation tiles: its latency is off the critical path of every wave but the last.
#define STAT_SHARDS 16
#define STAT_STRIDE 8                            /* uint64 per accumulator: 64 B apart */
#define STAT_WORDS ((STAT_SHARDS * 11 + 11) * STAT_STRIDE)
#define STAT_COUNT_BITS 17
#define STAT_FIX 65536.0                         /* 2^16: |sum| < 2^30 = 1e9 fits the 47-bit field (4 Mi envs x |term| <= 250) */
typedef GLOBAL_AS unsigned long long gu64;
struct StatsTicket { unsigned long long mine, old; };
DEV void stats_begin(const DevParams& P, const LaneStats& st, int lane, StatsTicket& tk) {
    float vals[11];
#pragma unroll
    for (int t = 0; t < 6; ++t) vals[t] = st.rew[t];
    vals[6] = st.pos_cnt; vals[7] = st.ori_cnt; vals[8] = st.succ; vals[9] = st.resets; vals[10] = st.nonfinite;
    float sum = 0.0f;                            // lane k < 11 ends up holding the wave's sum of statistic k
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const float s = wave_sum_lane63(vals[k]);
        const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), WAVE - 1));
        sum = (lane == k) ? b : sum;
    }
    const long long fx = (long long)((double)sum * STAT_FIX);
    tk.mine = ((unsigned long long)fx << STAT_COUNT_BITS) + 1ull;
    tk.old = 0ull;
    const int shard = (int)blockIdx.x & (STAT_SHARDS - 1);
    gu64* acc = (gu64*)P.tickets;
    if (lane < 11)
        tk.old = __hip_atomic_fetch_add(&acc[(shard * 11 + lane) * STAT_STRIDE], tk.mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEV void stats_end(const DevParams& P, int lane, const StatsTicket& tk) {
    const unsigned long long cmask = (1ull << STAT_COUNT_BITS) - 1ull;
    const int nw = (int)gridDim.x;
    const int shard = (int)blockIdx.x & (STAT_SHARDS - 1);
    const unsigned long long members = (unsigned long long)((nw - shard + STAT_SHARDS - 1) / STAT_SHARDS);
    const unsigned long long nshards = (unsigned long long)(nw < STAT_SHARDS ? nw : STAT_SHARDS);
    gu64* acc = (gu64*)P.tickets;
    if (lane < 11 && (tk.old & cmask) == members - 1ull) {          // last wave of this shard for statistic `lane`
        const unsigned long long total1 = tk.old + tk.mine;          // count field == members, sum field == shard sum
        __hip_atomic_store(&acc[(shard * 11 + lane) * STAT_STRIDE], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long v2 = (total1 & ~cmask) + 1ull;
        gu64* top = &acc[(STAT_SHARDS * 11 + lane) * STAT_STRIDE];
        const unsigned long long old2 = __hip_atomic_fetch_add(top, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old2 & cmask) == nshards - 1ull) {                     // last shard: the grand total is complete
            __hip_atomic_store(top, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long fxs = (long long)(old2 + v2) >> STAT_COUNT_BITS;
            const float total = (float)((double)fxs / STAT_FIX);
            const int k = lane;
            const float o = (k < 6 || k == 8) ? total / (float)P.N : total;
            const int slot = (k < 6) ? k : ((k == 6) ? TF_INFO_POS_COUNT : ((k == 7) ? TF_INFO_ORI_COUNT :
                             ((k == 8) ? TF_INFO_SUCCESS_MEAN : ((k == 9) ? TF_INFO_NUM_RESETS : TF_INFO_NUM_NONFINITE))));
            P.info[slot] = o;
        }
    }
}

DEV void stats_zero(LaneStats& st) {
#pragma unroll
    for (int t = 0; t < 6; ++t) st.rew[t] = 0.0f;
    st.pos_cnt = 0.0f; st.ori_cnt = 0.0f; st.succ = 0.0f; st.resets = 0.0f; st.nonfinite = 0.0f;
}


// arms of the three rows of a wall corner: r x n, r x t and (box only) r x z, inertia-scaled for a box (S from LDS)
DEV void wall_arms(bool box, const float* lds, int lane, const float r[3], const float n[2], float a[3], float b[3], float c3[3]) {
    wall_arm_n(r, n, a);
    wall_arm_t(r, n, b);
    c3[0] = 0.0f; c3[1] = 0.0f; c3[2] = 0.0f;
    if (__builtin_expect(box, 0)) {
        float S[6], t[3], ez[3];
#pragma unroll
        for (int e = 0; e < 6; ++e) S[e] = LD(L_POSE_S + e);
        sym3_mul(S, a, t); a[0] = t[0]; a[1] = t[1]; a[2] = t[2];
        sym3_mul(S, b, t); b[0] = t[0]; b[1] = t[1]; b[2] = t[2];
        box_axis(0, ez);
        box_arm(S, r, ez, c3);
    }
}

template <int A, bool IS_RESET, bool ASYM, int MODE, int X, bool WIDE, bool HELP = false>
DEV void cube_role(const DevParams& P, const StepArgs& sa, const float* __restrict__ action, float* lds, const Ctx& cx) {
    constexpr bool EXT = X != 0;
    constexpr bool BOXK = X == 2;
    const TfModel& m = P.m;
    const int lane = cx.lane;
    constexpr int OD = TF_OBS_DIM_BASE + A, SD = OD + TF_STATES_EXTRA;
    constexpr int TW = ASYM ? SD : OD;
    constexpr int NDR = EXT ? TF_NUM_DR : TF_DR_BASE_POS;
    const uint32_t gid = (uint32_t)(P.env_id_offset + cx.i);
    // ---- loads ----
    float cp[3], cq[4], cv[3], cw[3], dr[TF_NUM_DR];
    float lam_cf[12], lam_cw[12], cf_face = 0.0f, cw_face = 0.0f;
    uint8_t fl_reset = 0, fl_goal_reset = 0, fl_successes = 0;
    int fl_steps = 0;
    uint32_t fl_count = 0;
    STAMP(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) { cp[j] = LDST(TF_S_CUBE_P + j); cv[j] = LDST(TF_S_CUBE_V + j); cw[j] = LDST(TF_S_CUBE_W + j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) cq[j] = LDST(TF_S_CUBE_Q + j);
#pragma unroll
    for (int j = 0; j < TF_NUM_DR; ++j) dr[j] = (j < NDR && P.dr_enable) ? LDST(TF_S_DR + j) : TF_DR_NEUTRAL(j);   // rows are read only when the feature is on
    if (MODE & (M_RESETS | M_POST | M_FINISH)) {
        fl_reset = P.reset_buf[(unsigned)cx.i];
        fl_goal_reset = P.goal_reset_buf[(unsigned)cx.i];
        fl_successes = P.successes[(unsigned)cx.i];
        fl_steps = (int)P.steps[(unsigned)cx.i];
        fl_count = P.reset_count[(unsigned)cx.i];
    }
    if (MODE & M_ACT_RAND) draw_action_tile<A>(P, sa, lds, cx);
    else if (MODE & M_ACT_IN) coop_load_tile<A>(action, lds, cx);
    else if (MODE & (M_RESETS | M_TORQUE | M_POST)) coop_load_tile<A>((const float*)P.action_buf, lds, cx);
    if (P.dr_enable) {                                          // the env's domain-randomisation rows for the finger roles (L_DR0)
#pragma unroll
        for (int j = 0; j < NDR; ++j) LD(L_DR0 + j) = dr[j];
    }
    BAR();                                                      // #1
    if (!(MODE & (M_ACT_IN | M_RESETS))) BAR();                 // #1b: the finger roles have read L_DR0 (split path only; see the finger role)
    STAMP(1);
    if (MODE & (M_SIM | M_RESETS)) {                            // warm-start rows: behind the barrier, out of the launch's first load burst
        cf_face = LDST(TF_S_CF_FACE); cw_face = LDST(TF_S_CW_FACE);
#pragma unroll
        for (int j = 0; j < 12; ++j) { lam_cf[j] = LDST(TF_S_LAM_CF + j); lam_cw[j] = 0.0f; }
        // the wall-corner rows exist only while a corner touches the boundary (cw_face != 0): otherwise they are neither loaded nor stored
        const bool cw_was = cw_face != 0.0f;
        if (__builtin_amdgcn_ballot_w64(cw_was) != 0ull) {
#pragma unroll
            for (int j = 0; j < 12; ++j) { const float t = LDST(TF_S_LAM_CW + j); lam_cw[j] = cw_was ? t : 0.0f; }
        }
    }
    // flags carried in registers to the bookkeeping at the end of the step
    bool c_reset = fl_reset != 0, c_goal_reset = fl_goal_reset != 0, c_successes = fl_successes != 0;
    int c_steps = fl_steps;
    float n_resets = 0.0f;
    // ---- masked _reset_impl then _goal_reset_impl (env_base.py:370-379; trifinger_env.py:373-440) ----
    if (MODE & M_RESETS) {
        const bool rflag = IS_RESET || (fl_reset != 0);
        const bool gflag = !IS_RESET && (fl_goal_reset != 0);
        uint32_t count = fl_count;
        // the goal rows are only WRITTEN here (sample_goal fills every component; stored under rflag || gflag); the post phase reads them back
        float gp[3] = {0.0f, 0.0f, 0.0f}, gq[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gw[3] = {0.0f, 0.0f, 0.0f};
        // The samples of this reset were drawn at the END of the step that flagged it, by the finger wavefronts (TF_S_NEXT_*; the finger role's last
        // block): when the tag says they belong to this reset count they are loaded, and the ~600 instructions of Philox rounds, Box-Muller and
        // sine / cosine polynomials stay off the critical path of the launch (a launch ends with its slowest workgroup, and at any size some
        // workgroup holds a time-out: 1.4 us at 8192 envs, 2.5 us at 65536).  Same values either way: the draws are functions of (seed, env, count).
        float nx_obj[4] = {0.0f, 0.0f, 0.0f, 0.0f}, nx_goal[10] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        bool have = false;
        if (!IS_RESET && __builtin_amdgcn_ballot_w64(rflag) != 0ull) {      // wave-uniform: rows are read by the wavefronts that hold a reset
            const uint32_t tag = __builtin_bit_cast(uint32_t, LDST(TF_S_NEXT_TAG));
            have = rflag && (tag == count + 1u);
#pragma unroll
            for (int j = 0; j < 4; ++j) nx_obj[j] = LDST(TF_S_NEXT_OBJ + j);
#pragma unroll
            for (int j = 0; j < 10; ++j) nx_goal[j] = LDST(TF_S_NEXT_GOAL + j);
        }
        const bool draw_now = __builtin_amdgcn_ballot_w64(rflag && !have) != 0ull;      // wave-uniform: some lane's reset has no stored samples
        if (rflag) {
            if (P.dr_enable) draw_dr<EXT>(P, gid, count, dr);
            if (P.object_reset_type == TF_RESET_DEFAULT) {
                cp[0] = EXT ? 0.0f + dr[TF_DR_STAGE_POS] : 0.0f; cp[1] = EXT ? 0.0f + dr[TF_DR_STAGE_POS + 1] : 0.0f; cp[2] = m.obj_min_height * dr[1];
                cq[0] = 0.0f; cq[1] = 0.0f; cq[2] = 0.0f; cq[3] = 1.0f;
#pragma unroll
                for (int k = 0; k < 3; ++k) { cv[k] = 0.0f; cw[k] = 0.0f; }
            } else if (P.object_reset_type == TF_RESET_RANDOM) {
                float ox = nx_obj[0], oy = nx_obj[1], oq[4] = {0.0f, 0.0f, nx_obj[2], nx_obj[3]};
                if (draw_now) {
                    float u[4], tx, ty, tq[4];
                    rng4(P, gid, count, RNG_OBJECT, u);
                    sample_xy(u[0], u[1], m.obj_max_com_dist, tx, ty);
                    sample_yaw_quat(u[2], tq);
                    ox = have ? ox : tx; oy = have ? oy : ty; oq[2] = have ? oq[2] : tq[2]; oq[3] = have ? oq[3] : tq[3];
                }
                cp[0] = EXT ? ox + dr[TF_DR_STAGE_POS] : ox; cp[1] = EXT ? oy + dr[TF_DR_STAGE_POS + 1] : oy;      // spawn relative to the stage
                cp[2] = m.obj_min_height * dr[1];
#pragma unroll
                for (int k = 0; k < 4; ++k) cq[k] = oq[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) { cv[k] = 0.0f; cw[k] = 0.0f; }
            }
            if (draw_now) {
                float raw[10];
                sample_goal_pos(P, gid, count, &raw[0], &raw[3]);
                sample_goal_rot(P, gid, count, &raw[3], &raw[7]);
#pragma unroll
                for (int j = 0; j < 10; ++j) nx_goal[j] = have ? nx_goal[j] : raw[j];
            }
            goal_from_raw<EXT>(nx_goal, dr, gp, gq, gw);
            count = count + 1u;
#pragma unroll
            for (int j = 0; j < 12; ++j) { lam_cf[j] = 0.0f; lam_cw[j] = 0.0f; }
            cf_face = 0.0f; cw_face = 0.0f;
            n_resets = cx.valid ? 1.0f : 0.0f;
        }
        if (gflag) {
            sample_goal<EXT>(P, gid, count, dr, gp, gq, gw);
            count = count + 1u;
        }
        if (cx.valid) {
            if (rflag) { P.reset_buf[(unsigned)cx.i] = 0; P.steps[(unsigned)cx.i] = 0; P.successes[(unsigned)cx.i] = 0; }
            if (gflag) P.goal_reset_buf[(unsigned)cx.i] = 0;
            if (rflag || gflag) P.reset_count[(unsigned)cx.i] = count;
        }
        if (rflag && P.dr_enable) {
#pragma unroll
            for (int j = 0; j < NDR; ++j) STST(TF_S_DR + j, dr[j]);
        }
        if (rflag || gflag) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { STST(TF_S_GOAL_P + j, gp[j]); STST(TF_S_GOAL_W + j, gw[j]); }
#pragma unroll
            for (int j = 0; j < 4; ++j) STST(TF_S_GOAL_Q + j, gq[j]);
        }
        c_reset = false;                                        // cleared by the reset, or it was not set
        c_goal_reset = IS_RESET ? (fl_goal_reset != 0) : false; // reset() leaves _goal_reset_buf alone
        c_successes = rflag ? false : (fl_successes != 0);
        c_steps = rflag ? 0 : fl_steps;
    }
    if (MODE & (M_ACT_IN | M_RESETS)) {
        BAR();                                                  // #2a
        BAR();                                                  // #2b
        coop_store_tile<A>(P.action_buf, lds, cx);
    }
    if (MODE & M_TORQUE) {                                      // history[1] of the object (trifinger_env.py:975)
#pragma unroll
        for (int j = 0; j < 3; ++j) STST(TF_S_PREV_OBJ_P + j, cp[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) STST(TF_S_PREV_OBJ_Q + j, cq[j]);
    }
    // =================================================================================================================
    // physics
    // =================================================================================================================
    STAMP(2);
    if (MODE & M_SIM) {
        const float h = P.hsub, inv_h = 1.0f / h;
        const int nsub = sa.nsim * P.substeps;
        // The cube role carries the serial chain of the solve (every row depends on the one before it); the finger roles
        // of the other workgroups on this SIMD only fill its issue gaps.  Without a raised priority the older finger
        // wavefronts win the issue arbitration after every barrier and the chain waits for them (measured: the twelve
        // register-resident floor rows of a sweep took 3.1 k cycles instead of 0.8 k).
        __builtin_amdgcn_s_setprio(3);
        // the wall-corner impulses live in LDS through the substeps
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 0; d < 3; ++d) LD(L_WALL + 12 * c + 9 + d) = lam_cw[3 * c + d];
        }
        for (int s = 0; s < nsub; ++s) {
            const int sb_ = 4 + 12 * (s & 1);
            (void)sb_;
            float drs[TF_NUM_DR];                               // cold through the sweeps: re-read for every substep but the first
#pragma unroll
            for (int j = 0; j < TF_NUM_DR; ++j) drs[j] = (WIDE || j >= NDR || s == 0 || !P.dr_enable) ? dr[j] : LDST(TF_S_DR + j);
            const float* dr = drs;
            const float cube_mass = m.cube_mass * dr[0];
            const float cube_inertia = m.cube_inertia * dr[0] * dr[1] * dr[1];
            const float inv_m = 1.0f / cube_mass, inv_I = 1.0f / cube_inertia;
            const float fr1 = dr[TF_DR_FRICTION_ROBOT] - 1.0f, fo1 = dr[TF_DR_FRICTION_OBJECT] - 1.0f, fs1 = dr[TF_DR_FRICTION_STAGE] - 1.0f;
            const float mu_fc = (m.mu_finger_cube * dr[2]) * (EXT ? pair_factor(m.mu_robot, fr1, m.mu_object, fo1) : 1.0f);
            const float mu_cf = (m.mu_cube_floor * dr[2]) * (EXT ? pair_factor(m.mu_object, fo1, m.mu_floor, fs1) : 1.0f);
            const float mu_cw = (m.mu_cube_wall * dr[2]) * (EXT ? pair_factor(m.mu_object, fo1, m.mu_stage, fs1) : 1.0f);
            const float soff[2] = {dr[TF_DR_STAGE_POS], dr[TF_DR_STAGE_POS + 1]};
            const float rest_ff = m.restitution_ff * dr[5];
            const float ws = m.warm_start;
            constexpr bool box = BOXK;                        // general box object: its own kernel instantiations, the cube kernels carry none of it
            float hc[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) hc[j] = (box ? m.box_half[j] : m.cube_half) * dr[1];
            // ---- C1: free motion of the cube, corner contacts against the arena ----
            float v[3], w[3];
            float R[9];
            quat_to_rot(cq, R);
            {
                float wf[3] = {cw[0], cw[1], cw[2]};
                if (__builtin_expect(box && m.box_gyroscopic, 0)) {   // Euler's equations in the body frame, explicit
                    float wb[3], Iw[3], tq[3];
                    mat3T_mul(R, wf, wb);
#pragma unroll
                    for (int k = 0; k < 3; ++k) Iw[k] = m.box_inertia[k] * wb[k];
                    cross3(Iw, wb, tq);
#pragma unroll
                    for (int k = 0; k < 3; ++k) wb[k] = FMA(h, tq[k] / m.box_inertia[k], wb[k]);
                    mat3_mul(R, wb, wf);
                }
                float dl = 1.0f - h * m.cube_linear_damping, da = 1.0f - h * m.cube_angular_damping;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    v[j] = FMA(h, P.grav[j], cv[j]) * dl;
                    w[j] = wf[j] * da;
                }
                if (__builtin_expect(box, 0)) {                 // from here to the integration `w` is the inertia-scaled w^ = S^-1 w
                    float sc[3], si[3], S[6], Sinv[6], wh[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { sc[k] = f_sqrt(m.cube_inertia / m.box_inertia[k]); si[k] = 1.0f / sc[k]; }
                    rot_diag_rot(R, sc, S);
                    rot_diag_rot(R, si, Sinv);
                    sym3_mul(Sinv, w, wh);
#pragma unroll
                    for (int j = 0; j < 3; ++j) w[j] = wh[j];
#pragma unroll
                    for (int e = 0; e < 6; ++e) LD(L_POSE_S + e) = S[e];
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) { LD(L_POSE_A + j) = cp[j]; LD(L_POSE_B + j) = v[j]; LD(L_POSE_B + 3 + j) = w[j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) LD(L_POSE_A + 3 + j) = cq[j];
            // ---- corner contacts of the cube against the arena: built here, before S1, in the window in which the finger roles compute their free
            // motion and this role would only wait.  The 256-register build, whose fingers' free motion is short (one wavefront per SIMD), builds the
            // FLOOR corners behind S1b instead, in the window in which the finger roles build their records and this role waits again: there it was
            // the last to arrive at S1 in every substep (-1.0 us at 8192 envs; the 128-register build at 65536: +0.6 us, it keeps them here) ----
            float fr_[12], fDinv[12], fbias[4], flam[12];       // floor corners: arm, 1/D, bias, impulses of rows +z, +x, +y
            if constexpr (!WIDE) {
#include "tf_floor_corners.inc"
            }
            bool wall_lane = false;
            bool slot_any[4] = {false, false, false, false};     // wave-uniform: a lane of this wavefront has a live corner in slot c
            // WIDE (cube kernels): the rows of corner slots 0 and 1 stay in registers through the sweeps - normal, arms, 1/D, bias, impulses; LDS only
            // keeps their impulses between substeps (slots 2 and 3, practically never live, go through LDS as in the 128-register build)
            constexpr bool WREG = WIDE && !BOXK;
            float w_r[6], w_n[4], w_D[6], w_bias[2], w_lam[6], w_a[6], w_b[6];
            if constexpr (!HELP) {
#include "tf_wall_corners.inc"
            }
            // wave-uniform flags per corner slot: the sweeps enter the block of a slot only when a lane of this wavefront has a live corner
            // there - no LDS round trip for the others (with the lower pair in slots 0 and 1, slots 2 and 3 are practically never live)
            STAMP(sb_ + 0);
            BAR();                                              // S1
            STAMP(sb_ + 1);
            // ---- FF: finger-finger contacts (distal capsules), frictionless, resolved before the sweeps on the free
            // velocities: the pairs (0,1), (1,2), (2,0) in turn, one normal row each.  The velocities live in LDS (L_VQFF)
            // while the pairs are visited. ----
            if constexpr (HELP) {      // the distal pass runs on helper wavefront 7 (helper_role); this role builds the boundary corners in its place
#include "tf_wall_corners.inc"
            } else {
#include "tf_ff_distal.inc"
            }
            // ---- FF, second part (TfModel.ff_middle_pairs, on by default since API 8; wave-uniform): the middle link of finger fm (shape2) against the distal
            // capsule of each other finger - six ordered pairs on the same velocities.  The middle frame is rebuilt from what finger fm publishes:
            // e_x = (c1, 0, -s1), g = (p3 - p2) - j3_x e_x = j3_y e_y + j3_z e_z, e_x x g = j3_y e_z - j3_z e_y (oracle/tf_oracle.c, the same lines). ----
            // (placement per instantiation, same bits: the 256-register cube kernels build all six rows on the finger wavefronts - the block above
            // their S1b; the 128-register cube kernels leave the FIRST group, fd = fm + 2, here and build the second on the finger wavefronts; the box
            // kernels, whose LDS has no room for the mailboxes, visit both groups here)
            if (!WIDE && m.ff_middle_pairs != 0) {
                const TfLinkShape& sh = m.shape2;
                const float jx = m.j3_origin[0], jy = m.j3_origin[1], jz = m.j3_origin[2];
                const float inv_j = f_rcp(FMA(jy, jy, jz * jz));
#pragma unroll 1
                for (int o_ = 2; o_ >= (BOXK ? 1 : 2); --o_)
#pragma unroll 1
                for (int fm = 0; fm < 3; ++fm) {
                    FingerPubRegs pm;
                    read_pub(lds, lane, fm, pm);
                    const Yaw ym = {m.base_yaw_cos[fm], m.base_yaw_sin[fm], 0.0f, 0.0f, m.base_height};
                    const float ex[3] = {pm.k.c1, 0.0f, -pm.k.s1};
                    float ey[3], aw[3], bw[3];
                    {
                        float g[3], xg[3], ez[3], ab[3], bb[3];
#pragma unroll
                        for (int i = 0; i < 3; ++i) g[i] = FMA(-jx, ex[i], pm.k.p3[i] - pm.k.p2[i]);
                        cross3(ex, g, xg);
#pragma unroll
                        for (int i = 0; i < 3; ++i) { ey[i] = FMA(jy, g[i], -(jz * xg[i])) * inv_j; ez[i] = FMA(jz, g[i], jy * xg[i]) * inv_j; }
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            ab[i] = FMA(sh.a[2], ez[i], FMA(sh.a[1], ey[i], FMA(sh.a[0], ex[i], pm.k.p2[i])));
                            bb[i] = FMA(sh.b[2], ez[i], FMA(sh.b[1], ey[i], FMA(sh.b[0], ex[i], pm.k.p2[i])));
                        }
                        base_to_world(ym, ab, aw);
                        base_to_world(ym, bb, bw);
                    }
                    {
                        const int o = o_;
                        const int fd = (fm + o >= 3) ? fm + o - 3 : fm + o;
                        float Pm[3], Pd[3], sp;
                        {
                            float Ad[3], Bd[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) { Ad[j] = LD(L_REC(fd) + P_AW + j); Bd[j] = LD(L_REC(fd) + P_BW + j); }
                            seg_seg_s(aw, bw, Ad, Bd, Pm, Pd, sp);
                        }
                        const float dv[3] = {Pd[0] - Pm[0], Pd[1] - Pm[1], Pd[2] - Pm[2]};
                        const float dist2 = dot3(dv, dv);
                        const float inv = f_rsqrt(f_max(dist2, 1e-12f));
                        const float dist = dist2 * inv;
                        const float n[3] = {dv[0] * inv, dv[1] * inv, dv[2] * inv};      // from the middle link to the distal capsule
                        float nm[3];
                        dir_world_to_base(ym, n, nm);
                        const float u1 = dot3(nm, ex), u2 = dot3(nm, ey);
                        const float rho = FMA(sp, sh.rho[1] - sh.rho[0], sh.rho[0]);
                        const float h1 = FMA(sp, sh.w1[1] - sh.w1[0], sh.w1[0]) - rho, h2 = FMA(sp, sh.w2[1] - sh.w2[0], sh.w2[0]) - rho;
                        const float o1 = FMA(sp, sh.o1[1] - sh.o1[0], sh.o1[0]), o2 = FMA(sp, sh.o2[1] - sh.o2[0], sh.o2[0]);
                        const float ext = FMA(o2, u2, FMA(o1, u1, FMA(h2, f_abs(u2), FMA(h1, f_abs(u1), rho))));
                        const float gap = dist - ext - m.cap_radius;
                        if ((dist2 > 1e-12f) && (gap < m.contact_margin)) {
                            float Jd[3], Wd[3], Jm[3], Wm[3], vd[3], vm[3];
                            {   // distal side: the point of the capsule surface that faces the middle link
                                const Yaw yd = {m.base_yaw_cos[fd], m.base_yaw_sin[fd], 0.0f, 0.0f, m.base_height};
                                FingerPubRegs pd;
                                read_pub(lds, lane, fd, pd);
                                float C[3], Cb_[3], L1[3], L2[3], L3[3], nb[3];
#pragma unroll
                                for (int j = 0; j < 3; ++j) C[j] = FMA(-m.cap_radius, n[j], Pd[j]);
                                world_to_base(yd, C, Cb_);
                                levers(pd.k, Cb_, L1, L2, L3);
                                dir_world_to_base(yd, n, nb);
                                Jd[0] = dot3(L1, nb); Jd[1] = dot3(L2, nb); Jd[2] = dot3(L3, nb);
                                sym3_mul(pd.k.Minv, Jd, Wd);
                            }
                            {   // middle side: joints 1 and 2 move it, joint 3 does not
                                float C[3], Cb_[3], L1[3], L2[3], L3[3];
#pragma unroll
                                for (int j = 0; j < 3; ++j) C[j] = FMA(ext, n[j], Pm[j]);
                                world_to_base(ym, C, Cb_);
                                levers(pm.k, Cb_, L1, L2, L3);
                                Jm[0] = dot3(L1, nm); Jm[1] = dot3(L2, nm); Jm[2] = 0.0f;
                                sym3_mul(pm.k.Minv, Jm, Wm);
                            }
#pragma unroll
                            for (int j = 0; j < 3; ++j) { vd[j] = LD(L_REC(fd) + P_VQ + j); vm[j] = LD(L_REC(fm) + P_VQ + j); }      // the FREE velocities (API 8)
                            const float vn0 = dot3(Jd, vd) - dot3(Jm, vm);
                            if (contact_live(m, gap, vn0, h)) {
                                const float bias = contact_bias(m, gap, vn0, inv_h, rest_ff);
                                const float lam = f_max(-(vn0 + bias) * f_rcp2(dot3(Jd, Wd) + dot3(Jm, Wm)), 0.0f);
#pragma unroll
                                for (int j = 0; j < 3; ++j) {
                                    const float dd = Wd[j] * lam, dm = Wm[j] * lam;
                                    if (dd != 0.0f) LD(L_VQFF + 3 * fd + j) = LD(L_VQFF + 3 * fd + j) + dd;
                                    if (dm != 0.0f) LD(L_VQFF + 3 * fm + j) = LD(L_VQFF + 3 * fm + j) - dm;
                                }
                            }
                        }
                    }
                }
            }
            STAMP(sb_ + 2);
            BAR();                                              // S1b: finger-finger pass done
            STAMP(sb_ + 3);
            if constexpr (WIDE) {
#include "tf_floor_corners.inc"
            }
            STAMP(sb_ + 4);
            BAR();                                              // S3: records published by the finger roles
            STAMP(sb_ + 5);
            if constexpr (WIDE && !BOXK) {
            // ---- the 256-register instantiation of the cube kernels: contact-space records as (direction, arm) register pairs read once per
            // substep, the twist as three (v_j, w_j) pairs, the general rows as packed fp32 (tf_contact.h: the same lane-operations in the
            // same order as the form below); with one wavefront per SIMD the sweep is bound by issue slots, and this is half of them ----
            Twist tw;
#pragma unroll
            for (int j = 0; j < 3; ++j) { tw.p[j].x = v[j]; tw.p[j].y = w[j]; }
            float2v mI; mI.x = inv_m; mI.y = inv_I;
            float cDinv[9], cbias[3], clam[9], Km[9];
            float2v rec[27];                                    // rec[9 f + 3 d + j] = (dir_d[j], (r x dir_d)[j]) of finger f (a dead slot's record is stale LDS content:
#pragma unroll                                                  //  loaded, never used - its rows sit behind the 1/D > 0 branch)
            for (int f = 0; f < 3; ++f) {
                const int rb = L_REC(f);
                cbias[f] = LD(L_INIT + f);
#pragma unroll
                for (int d = 0; d < 3; ++d) { cDinv[3 * f + d] = LD(L_VQFF + 3 * f + d); clam[3 * f + d] = LD(rb + R_DL + d); Km[3 * f + d] = LD(rb + R_A + d); }
#pragma unroll
                for (int j = 0; j < 9; ++j) { rec[9 * f + j].x = LD(rb + R_DIR + j); rec[9 * f + j].y = LD(rb + R_RXD + j); }
                if (cDinv[3 * f] > 0.0f) {                       // seeded impulses of a live contact
#pragma unroll
                    for (int d = 0; d < 3; ++d) pk_row_apply_neg(&rec[9 * f + 3 * d], clam[3 * f + d], mI, tw);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (fDinv[3 * c] > 0.0f) {
                    cz_apply(&fr_[3 * c], flam[3 * c], mI, tw);
                    cx_apply(&fr_[3 * c], flam[3 * c + 1], mI, tw);
                    cy_apply(&fr_[3 * c], flam[3 * c + 2], mI, tw);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int wb = L_WALL + 12 * c;
                if (c < 2) {
                    if (slot_any[c] && w_D[3 * c] > 0.0f) {
                        wn_apply(&w_n[2 * c], &w_a[3 * c], w_lam[3 * c], mI, tw);
                        wt_apply(&w_n[2 * c], &w_b[3 * c], w_lam[3 * c + 1], mI, tw);
                        cz_apply(&w_r[3 * c], w_lam[3 * c + 2], mI, tw);
                    }
                } else if (slot_any[c] && LD(wb + 5) > 0.0f) {
                    float r[3] = {LD(wb), LD(wb + 1), LD(wb + 2)}, n[2] = {LD(wb + 3), LD(wb + 4)};
                    float a[3], b[3];
                    wall_arm_n(r, n, a);
                    wall_arm_t(r, n, b);
                    wn_apply(n, a, LD(wb + 9), mI, tw);
                    wt_apply(n, b, LD(wb + 10), mI, tw);
                    cz_apply(r, LD(wb + 11), mI, tw);
                }
            }
            uint32_t t_wait = 0u, t_fc = 0u, t_floor = 0u;
            STAMP(sb_ + 9);
            for (int it = 0; it < P.iters; ++it) {
                const uint32_t tf0_ = NOW();
                const bool last = it == P.iters - 1;
                float uu[9];                                    // the nine contact-point velocities of the sweep, requested together
#pragma unroll
                for (int f = 0; f < 3; ++f) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) uu[3 * f + j] = LD(L_REC(f) + R_U + j);
                }
#pragma unroll
                for (int f = 0; f < 3; ++f) {                   // finger-cube rows in contact space (block form with K: see the general path below)
#ifdef TF_DIAG_NO_FC
                    if (false) {
#else
                    if (cDinv[3 * f] > 0.0f) {
#endif
                        const int rb = L_REC(f);
                        const float2v* R = &rec[9 * f];
                        float vr[3], dl[3];
                        {   // the three rows side by side (no packed instruction waits for the one before it)
                            float2v p0 = R[0] * tw.p[0], p1 = R[3] * tw.p[0], p2 = R[6] * tw.p[0];
                            p0 = pk_fma(R[1], tw.p[1], p0); p1 = pk_fma(R[4], tw.p[1], p1); p2 = pk_fma(R[7], tw.p[1], p2);
                            p0 = pk_fma(R[2], tw.p[2], p0); p1 = pk_fma(R[5], tw.p[2], p1); p2 = pk_fma(R[8], tw.p[2], p2);
                            vr[0] = uu[3 * f] - (p0.x + p0.y); vr[1] = uu[3 * f + 1] - (p1.x + p1.y); vr[2] = uu[3 * f + 2] - (p2.x + p2.y);
                        }
                        dl[0] = solve_normal(clam[3 * f], cDinv[3 * f], vr[0], cbias[f]);
                        vr[1] = FMA(Km[3 * f], dl[0], vr[1]);
                        dl[1] = solve_tangent(clam[3 * f + 1], cDinv[3 * f + 1], vr[1], mu_fc * clam[3 * f]);
                        vr[2] = FMA(Km[3 * f + 2], dl[1], FMA(Km[3 * f + 1], dl[0], vr[2]));
                        dl[2] = solve_tangent(clam[3 * f + 2], cDinv[3 * f + 2], vr[2], mu_fc * clam[3 * f]);
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            LD(rb + R_DL + d) = dl[d];
                            pk_row_apply_neg(&R[3 * d], dl[d], mI, tw);
                        }
                    }
                }
                t_fc += NOW() - tf0_;
                { const uint32_t t0_ = NOW(); BAR(); t_wait += NOW() - t0_; }   // W1
#pragma unroll
                for (int c = 0; c < 4; ++c) {                   // cube - floor: rows +z (normal), +x, +y
                    const float* r = &fr_[3 * c];
#ifdef TF_DIAG_NO_FLOOR
                    if (false) {
#else
                    if (fDinv[3 * c] > 0.0f) {
#endif
                        float dl = solve_normal(flam[3 * c], fDinv[3 * c], cz_vrel(r, tw), fbias[c]);
                        cz_apply(r, dl, mI, tw);
                        dl = solve_tangent(flam[3 * c + 1], fDinv[3 * c + 1], cx_vrel(r, tw), mu_cf * flam[3 * c]);
                        cx_apply(r, dl, mI, tw);
                        dl = solve_tangent(flam[3 * c + 2], fDinv[3 * c + 2], cy_vrel(r, tw), mu_cf * flam[3 * c]);
                        cy_apply(r, dl, mI, tw);
                    }
                }
                t_floor += NOW() - tf0_;
#pragma unroll
                for (int c = 0; c < 4; ++c) {                   // cube - wall: rows n (normal), t, +z (dead lanes: n = 0, 1/D = 0, zero impulses; see below)
#ifdef TF_DIAG_NO_WALL
                    continue;
#endif
                    if (!slot_any[c]) continue;                 // wave-uniform
                    const int wb = L_WALL + 12 * c;
                    if (c < 2) {
                        float dl = solve_normal(w_lam[3 * c], w_D[3 * c], wn_vrel(&w_n[2 * c], &w_a[3 * c], tw), w_bias[c]);
                        wn_apply(&w_n[2 * c], &w_a[3 * c], dl, mI, tw);
                        dl = solve_tangent(w_lam[3 * c + 1], w_D[3 * c + 1], wt_vrel(&w_n[2 * c], &w_b[3 * c], tw), mu_cw * w_lam[3 * c]);
                        wt_apply(&w_n[2 * c], &w_b[3 * c], dl, mI, tw);
                        dl = solve_tangent(w_lam[3 * c + 2], w_D[3 * c + 2], cz_vrel(&w_r[3 * c], tw), mu_cw * w_lam[3 * c]);
                        cz_apply(&w_r[3 * c], dl, mI, tw);
                        if (last && w_D[3 * c] > 0.0f) {
#pragma unroll
                            for (int d = 0; d < 3; ++d) LD(wb + 9 + d) = w_lam[3 * c + d];
                        }
                    } else {
                        float r[3] = {LD(wb), LD(wb + 1), LD(wb + 2)}, n[2] = {LD(wb + 3), LD(wb + 4)};
                        float Dinv[3] = {LD(wb + 5), LD(wb + 6), LD(wb + 7)}, bias = LD(wb + 8);
                        float lam[3] = {LD(wb + 9), LD(wb + 10), LD(wb + 11)};
                        float a[3], b[3];
                        wall_arm_n(r, n, a);
                        wall_arm_t(r, n, b);
                        float dl = solve_normal(lam[0], Dinv[0], wn_vrel(n, a, tw), bias);
                        wn_apply(n, a, dl, mI, tw);
                        dl = solve_tangent(lam[1], Dinv[1], wt_vrel(n, b, tw), mu_cw * lam[0]);
                        wt_apply(n, b, dl, mI, tw);
                        dl = solve_tangent(lam[2], Dinv[2], cz_vrel(r, tw), mu_cw * lam[0]);
                        cz_apply(r, dl, mI, tw);
                        if (Dinv[0] > 0.0f) {
#pragma unroll
                            for (int d = 0; d < 3; ++d) LD(wb + 9 + d) = lam[d];
                        }
                    }
                }
                if (last) {                                     // what the finger roles keep: normal impulse, world friction impulse, force (read behind W2)
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const int rb = L_REC(f);
                        float ftv[3] = {0.0f, 0.0f, 0.0f}, Fc[3] = {0.0f, 0.0f, 0.0f};
                        if (cDinv[3 * f] > 0.0f) {
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                ftv[j] = FMA(rec[9 * f + 6 + j].x, clam[3 * f + 2], rec[9 * f + 3 + j].x * clam[3 * f + 1]);
                                Fc[j] = FMA(rec[9 * f + j].x, clam[3 * f], ftv[j]) * inv_h;
                            }
                        }
                        LD(L_INIT + f) = clam[3 * f];
#pragma unroll
                        for (int j = 0; j < 3; ++j) { LD(rb + R_A + j) = ftv[j]; LD(rb + R_A + 3 + j) = Fc[j]; }
                    }
                }
                { const uint32_t t0_ = NOW(); BAR(); t_wait += NOW() - t0_; }   // W2
            }
            STAMP(sb_ + 6);
            STAMPV(sb_ + 8, t_wait);
            STAMPV(sb_ + 10, t_fc);
            STAMPV(sb_ + 11, t_floor);
#pragma unroll
            for (int j = 0; j < 3; ++j) { v[j] = tw.p[j].x; w[j] = tw.p[j].y; }
            } else {
            // ---- seeded impulses of the finger contacts (1/D, bias and impulses stay in registers through the sweeps) ----
            float cDinv[9], cbias[3], clam[9];
            // WIDE: the contact-space records (A, directions, arms) are read ONCE per substep and stay in registers through the sweeps; only the
            // contact-point velocity u, which the finger role republishes in every sweep, still comes through LDS (a dead slot's record is stale
            // LDS content: loaded, never used - its rows sit behind the 1/D > 0 branch)
            constexpr int NRR = WIDE ? 27 : 1, NRA = WIDE ? 9 : 1;
            float rA[NRA], rDir[NRR], rRxd[NRR];
            if (WIDE) {
#pragma unroll
                for (int f = 0; f < 3; ++f) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) rA[(3 * f + j) % NRA] = LD(L_REC(f) + R_A + j);
#pragma unroll
                    for (int j = 0; j < 9; ++j) { rDir[(9 * f + j) % NRR] = LD(L_REC(f) + R_DIR + j); rRxd[(9 * f + j) % NRR] = LD(L_REC(f) + R_RXD + j); }
                }
            }
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int rb = L_REC(f);
                cbias[f] = LD(L_INIT + f);
#pragma unroll
                for (int d = 0; d < 3; ++d) { cDinv[3 * f + d] = LD(L_VQFF + 3 * f + d); clam[3 * f + d] = LD(rb + R_DL + d); }
                if (cDinv[3 * f] > 0.0f) {                       // a live contact has 1/D > 0
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float dir[3], rxd[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) { dir[j] = WIDE ? rDir[(9 * f + 3 * d + j) % NRR] : LD(rb + R_DIR + 3 * d + j); rxd[j] = WIDE ? rRxd[(9 * f + 3 * d + j) % NRR] : LD(rb + R_RXD + 3 * d + j); }
                        float sc = clam[3 * f + d] * inv_m, qq = clam[3 * f + d] * inv_I;
#pragma unroll
                        for (int j = 0; j < 3; ++j) { v[j] = FMA(-dir[j], sc, v[j]); w[j] = FMA(-rxd[j], qq, w[j]); }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (fDinv[3 * c] > 0.0f) {
                    if (__builtin_expect(box, 0)) {
                        float S[6];
#pragma unroll
                        for (int e = 0; e < 6; ++e) S[e] = LD(L_POSE_S + e);
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            float n[3], a[3];
                            box_axis(d, n);
                            box_arm(S, &fr_[3 * c], n, a);
                            g_apply(n, a, flam[3 * c + d], inv_m, inv_I, v, w);
                        }
                    } else {
                        cz_apply(&fr_[3 * c], flam[3 * c], inv_m, inv_I, v, w);
                        cx_apply(&fr_[3 * c], flam[3 * c + 1], inv_m, inv_I, v, w);
                        cy_apply(&fr_[3 * c], flam[3 * c + 2], inv_m, inv_I, v, w);
                    }
                }
            }
            {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int wb = L_WALL + 12 * c;
                if (WREG && c < 2) {
                    if (slot_any[c] && w_D[3 * c] > 0.0f) {
                        wn_apply(&w_n[2 * c], &w_a[3 * c], w_lam[3 * c], inv_m, inv_I, v, w);
                        wt_apply(&w_n[2 * c], &w_b[3 * c], w_lam[3 * c + 1], inv_m, inv_I, v, w);
                        cz_apply(&w_r[3 * c], w_lam[3 * c + 2], inv_m, inv_I, v, w);
                    }
                } else
                if (slot_any[c] && LD(wb + 5) > 0.0f) {
                    float r[3] = {LD(wb), LD(wb + 1), LD(wb + 2)}, n[2] = {LD(wb + 3), LD(wb + 4)};
                    float a[3], b[3], c3[3];
                    wall_arms(box, lds, lane, r, n, a, b, c3);
                    wn_apply(n, a, LD(wb + 9), inv_m, inv_I, v, w);
                    wt_apply(n, b, LD(wb + 10), inv_m, inv_I, v, w);
                    if (__builtin_expect(box, 0)) { float ez[3]; box_axis(0, ez); g_apply(ez, c3, LD(wb + 11), inv_m, inv_I, v, w); }
                    else cz_apply(r, LD(wb + 11), inv_m, inv_I, v, w);
                }
            }
            }
            // ---- projected Gauss-Seidel: the cube role's share ----
            uint32_t t_wait = 0u, t_fc = 0u, t_floor = 0u;
            STAMP(sb_ + 9);
            for (int it = 0; it < P.iters; ++it) {
                const uint32_t tf0_ = NOW();
                const bool last = it == P.iters - 1;
                float uu[9];                                    // WIDE: the nine contact-point velocities of the sweep, requested together
                if (WIDE) {
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) uu[3 * f + j] = LD(L_REC(f) + R_U + j);
                    }
                }
#pragma unroll
                for (int f = 0; f < 3; ++f) {                   // finger-cube rows in contact space
                    const int rb = L_REC(f);
                    if (cDinv[3 * f] > 0.0f) {
                        // The three rows of the block from the relative velocities at the INCOMING twist (independent of each other), the later
                        // rows corrected with the off-diagonal entries of the block-local Delassus matrix K: the iterate of the row-by-row
                        // Gauss-Seidel form in exact arithmetic, with the twist updates off the dependency chain (oracle: same order).
                        float Km[3], u[3], dirs[9], rxds[9], vr[3], dl[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) { Km[j] = WIDE ? rA[(3 * f + j) % NRA] : LD(rb + R_A + j); u[j] = WIDE ? uu[3 * f + j] : LD(rb + R_U + j); }
#pragma unroll
                        for (int j = 0; j < 9; ++j) { dirs[j] = WIDE ? rDir[(9 * f + j) % NRR] : LD(rb + R_DIR + j); rxds[j] = WIDE ? rRxd[(9 * f + j) % NRR] : LD(rb + R_RXD + j); }
#pragma unroll
                        for (int d = 0; d < 3; ++d) vr[d] = u[d] - (dot3(&dirs[3 * d], v) + dot3(&rxds[3 * d], w));
                        dl[0] = solve_normal(clam[3 * f], cDinv[3 * f], vr[0], cbias[f]);
                        vr[1] = FMA(Km[0], dl[0], vr[1]);
                        dl[1] = solve_tangent(clam[3 * f + 1], cDinv[3 * f + 1], vr[1], mu_fc * clam[3 * f]);
                        vr[2] = FMA(Km[2], dl[1], FMA(Km[1], dl[0], vr[2]));
                        dl[2] = solve_tangent(clam[3 * f + 2], cDinv[3 * f + 2], vr[2], mu_fc * clam[3 * f]);
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            LD(rb + R_DL + d) = dl[d];
                            const float sc = dl[d] * inv_m, qq = dl[d] * inv_I;
#pragma unroll
                            for (int j = 0; j < 3; ++j) { v[j] = FMA(-dirs[3 * d + j], sc, v[j]); w[j] = FMA(-rxds[3 * d + j], qq, w[j]); }
                        }
                    }
                }
                t_fc += NOW() - tf0_;
                { const uint32_t t0_ = NOW(); BAR(); t_wait += NOW() - t0_; }   // W1
#pragma unroll
                for (int c = 0; c < 4; ++c) {                   // cube - floor: rows +z (normal), +x, +y
                    const float* r = &fr_[3 * c];
                    if (__builtin_expect(box, 0)) {
                        if (fDinv[3 * c] > 0.0f) {
                            float S[6];
#pragma unroll
                            for (int e = 0; e < 6; ++e) S[e] = LD(L_POSE_S + e);
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                float n[3], a[3];
                                box_axis(d, n);
                                box_arm(S, r, n, a);
                                const float vrel = g_vrel(n, a, v, w);
                                const float dlb = (d == 0) ? solve_normal(flam[3 * c], fDinv[3 * c], vrel, fbias[c])
                                                           : solve_tangent(flam[3 * c + d], fDinv[3 * c + d], vrel, mu_cf * flam[3 * c]);
                                g_apply(n, a, dlb, inv_m, inv_I, v, w);
                            }
                        }
                    } else
                    if (fDinv[3 * c] > 0.0f) {
                        float dl = solve_normal(flam[3 * c], fDinv[3 * c], cz_vrel(r, v, w), fbias[c]);
                        cz_apply(r, dl, inv_m, inv_I, v, w);
                        dl = solve_tangent(flam[3 * c + 1], fDinv[3 * c + 1], cx_vrel(r, v, w), mu_cf * flam[3 * c]);
                        cx_apply(r, dl, inv_m, inv_I, v, w);
                        dl = solve_tangent(flam[3 * c + 2], fDinv[3 * c + 2], cy_vrel(r, v, w), mu_cf * flam[3 * c]);
                        cy_apply(r, dl, inv_m, inv_I, v, w);
                    }
                }
                t_floor += NOW() - tf0_;
                {
#pragma unroll
                for (int c = 0; c < 4; ++c) {                   // cube - wall: rows n (normal), t, +z
                    if (!slot_any[c]) continue;                 // wave-uniform
                    const int wb = L_WALL + 12 * c;
                    if (WREG && c < 2) {                        // the same rows on the register copies (dead lanes: n = 0, 1/D = 0, zero impulses, as below)
                        float dl = solve_normal(w_lam[3 * c], w_D[3 * c], wn_vrel(&w_n[2 * c], &w_a[3 * c], v, w), w_bias[c]);
                        wn_apply(&w_n[2 * c], &w_a[3 * c], dl, inv_m, inv_I, v, w);
                        dl = solve_tangent(w_lam[3 * c + 1], w_D[3 * c + 1], wt_vrel(&w_n[2 * c], &w_b[3 * c], v, w), mu_cw * w_lam[3 * c]);
                        wt_apply(&w_n[2 * c], &w_b[3 * c], dl, inv_m, inv_I, v, w);
                        dl = solve_tangent(w_lam[3 * c + 2], w_D[3 * c + 2], cz_vrel(&w_r[3 * c], v, w), mu_cw * w_lam[3 * c]);
                        cz_apply(&w_r[3 * c], dl, inv_m, inv_I, v, w);
                        if (last && w_D[3 * c] > 0.0f) {
#pragma unroll
                            for (int d = 0; d < 3; ++d) LD(wb + 9 + d) = w_lam[3 * c + d];
                        }
                    } else {   // No per-lane branch around the rows: a lane whose corner in this slot is not live holds n = 0, 1/D = 0, bias = 0 and
                        // zero impulses there, so its rows come out as dl = +-0 and leave v, w as they are (they are never -0: they start at +0
                        // and only ever pass through additions) - one LDS round trip per block instead of two, no exec-mask juggling; only the
                        // impulses are stored per lane (a clamp to +-0 can yield -0, which must not reach the state row of a dead slot).
                        float r[3] = {LD(wb), LD(wb + 1), LD(wb + 2)}, n[2] = {LD(wb + 3), LD(wb + 4)};
                        float Dinv[3] = {LD(wb + 5), LD(wb + 6), LD(wb + 7)}, bias = LD(wb + 8);
                        float lam[3] = {LD(wb + 9), LD(wb + 10), LD(wb + 11)};
                        float a[3], b[3], c3[3];
                        wall_arms(box, lds, lane, r, n, a, b, c3);
                        float dl = solve_normal(lam[0], Dinv[0], wn_vrel(n, a, v, w), bias);
                        wn_apply(n, a, dl, inv_m, inv_I, v, w);
                        dl = solve_tangent(lam[1], Dinv[1], wt_vrel(n, b, v, w), mu_cw * lam[0]);
                        wt_apply(n, b, dl, inv_m, inv_I, v, w);
                        if (__builtin_expect(box, 0)) {
                            float ez[3];
                            box_axis(0, ez);
                            dl = solve_tangent(lam[2], Dinv[2], g_vrel(ez, c3, v, w), mu_cw * lam[0]);
                            g_apply(ez, c3, dl, inv_m, inv_I, v, w);
                        } else {
                            dl = solve_tangent(lam[2], Dinv[2], cz_vrel(r, v, w), mu_cw * lam[0]);
                            cz_apply(r, dl, inv_m, inv_I, v, w);
                        }
                        if (Dinv[0] > 0.0f) {
#pragma unroll
                            for (int d = 0; d < 3; ++d) LD(wb + 9 + d) = lam[d];
                        }
                    }
                }
                }
                if (last) {                                     // what the finger roles keep: normal impulse, world friction impulse, force (read behind W2)
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const int rb = L_REC(f);
                        float ftv[3] = {0.0f, 0.0f, 0.0f}, Fc[3] = {0.0f, 0.0f, 0.0f};
                        if (cDinv[3 * f] > 0.0f) {
                            float dirs[9];
#pragma unroll
                            for (int j = 0; j < 9; ++j) dirs[j] = WIDE ? rDir[(9 * f + j) % NRR] : LD(rb + R_DIR + j);
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                ftv[j] = FMA(dirs[6 + j], clam[3 * f + 2], dirs[3 + j] * clam[3 * f + 1]);
                                Fc[j] = FMA(dirs[j], clam[3 * f], ftv[j]) * inv_h;
                            }
                        }
                        LD(L_INIT + f) = clam[3 * f];
#pragma unroll
                        for (int j = 0; j < 3; ++j) { LD(rb + R_A + j) = ftv[j]; LD(rb + R_A + 3 + j) = Fc[j]; }
                    }
                }
                { const uint32_t t0_ = NOW(); BAR(); t_wait += NOW() - t0_; }   // W2
            }
            STAMP(sb_ + 6);
            STAMPV(sb_ + 8, t_wait);
            STAMPV(sb_ + 10, t_fc);
            STAMPV(sb_ + 11, t_floor);
            }
            // ---- impulses kept for the next substep, integration ----
#pragma unroll
            for (int j = 0; j < 12; ++j) lam_cf[j] = flam[j];
            if (__builtin_expect(box, 0)) {                     // back to the world angular velocity: w = S w^
                float S[6], ww[3];
#pragma unroll
                for (int e = 0; e < 6; ++e) S[e] = LD(L_POSE_S + e);
                sym3_mul(S, w, ww);
#pragma unroll
                for (int j = 0; j < 3; ++j) w[j] = ww[j];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                cv[j] = v[j]; cw[j] = w[j];
                cp[j] = FMA(h, v[j], cp[j]);
            }
            quat_integrate(cq, cw, h);
            STAMP(sb_ + 7);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 0; d < 3; ++d) lam_cw[3 * c + d] = LD(L_WALL + 12 * c + 9 + d);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    STAMP(30);
    // =================================================================================================================
    // post: observations of the object and the goal, rewards, termination, statistics
    // =================================================================================================================
    StatsTicket tk;
    tk.mine = 0ull; tk.old = 0ull;
    if (MODE & (M_POST | M_FINISH)) {     // the flag buffers as the reset logic left them (cold through the physics: re-read)
        c_reset = P.reset_buf[(unsigned)cx.i] != 0;
        c_goal_reset = P.goal_reset_buf[(unsigned)cx.i] != 0;
        c_successes = P.successes[(unsigned)cx.i] != 0;
        c_steps = (int)P.steps[(unsigned)cx.i];
    }
    if (MODE & M_POST) {
        float gp[3], gq[4], prev_obj[7];                        // cold through the physics: re-read from their rows
#pragma unroll
        for (int j = 0; j < 3; ++j) gp[j] = LDST(TF_S_GOAL_P + j);
#pragma unroll
        for (int j = 0; j < 4; ++j) gq[j] = LDST(TF_S_GOAL_Q + j);
#pragma unroll
        for (int j = 0; j < 7; ++j) prev_obj[j] = LDST(TF_S_PREV_OBJ_P + j);
        LaneStats st;
        stats_zero(st);
        st.resets = n_resets;
        {   // NaN guard, part 1: this role's share of the finiteness test
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = acc + cp[j] * 0.0f + cv[j] * 0.0f + cw[j] * 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = acc + cq[j] * 0.0f;
            LD(L_NAN + 3) = (acc == 0.0f) ? 0.0f : 1.0f;
        }
        // The object terms of the reward (distance kernel, rotation terms, object motion: everything that needs no fingertip) depend on what this role
        // already holds.  They are evaluated HERE, in the window in which this role otherwise waits for the finger roles at P1 and P3; the 256-register
        // instantiation carries the six results in registers, the 128-register one parks them in LDS (L_OTERM).
        float o_dist = 0.0f, o_ang = 0.0f, o_r[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        auto object_terms = [&]() __attribute__((always_inline)) {
            const RewardCoef& rc = sa.rc;
            o_dist = norm3d(cp, gp);
            o_r[0] = rc.c_dist * lgsk(o_dist, 50.0f);
            o_ang = quat_diff_rad(cq, gq);
            o_r[1] = rc.w_rot * (rc.rot_num / (rc.rot_scale * f_abs(o_ang) + rc.rot_scale));
            const float ang_prev = quat_diff_rad(&prev_obj[3], gq);
            o_r[2] = rc.w_rot_delta * (rc.rot_delta_sched * (f_abs(o_ang) - f_abs(ang_prev)));
            o_r[3] = rc.w_move * (o_dist - norm3d(prev_obj, gp));
        };
        if (!IS_RESET) {
            object_terms();
            if (!WIDE) { LD(L_OTERM) = o_dist; LD(L_OTERM + 1) = o_ang; LD(L_OTERM + 2) = o_r[0]; LD(L_OTERM + 3) = o_r[1]; LD(L_OTERM + 4) = o_r[2]; LD(L_OTERM + 5) = o_r[3]; }
        }
        STAMP(31);
        BAR();                                                  // P1
        STAMP(32);
        const bool guarded = (LD(L_NAN) + LD(L_NAN + 1) + LD(L_NAN + 2) + LD(L_NAN + 3)) != 0.0f;
        if (__builtin_expect(guarded, 0)) {     // a non-finite env is flagged for reset and parked at the default pose
            cp[0] = 0.0f; cp[1] = 0.0f; cp[2] = m.obj_min_height;
            cq[0] = 0.0f; cq[1] = 0.0f; cq[2] = 0.0f; cq[3] = 1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { cv[k] = 0.0f; cw[k] = 0.0f; }
            if (cx.valid) P.reset_buf[(unsigned)cx.i] = 1;
            c_reset = true;
            st.nonfinite += cx.valid ? 1.0f : 0.0f;
            if (!IS_RESET) {                                  // (the terms of the parked pose: what an evaluation behind P3 would see)
                object_terms();
                if (!WIDE) { LD(L_OTERM) = o_dist; LD(L_OTERM + 1) = o_ang; LD(L_OTERM + 2) = o_r[0]; LD(L_OTERM + 3) = o_r[1]; LD(L_OTERM + 4) = o_r[2]; LD(L_OTERM + 5) = o_r[3]; }
            }
        }
        const float co = P.clip_obs;
        const bool nrm = P.normalize_obs != 0;
        float* row = &lds[lane * TW];
#pragma unroll
        for (int j = 0; j < 3; ++j) { row[18 + j] = emit_scaled(cp[j], PLO(j), 0.3f, nrm, co); row[25 + j] = emit_scaled(gp[j], PLO(j), 0.3f, nrm, co); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { row[21 + j] = emit_scaled(cq[j], -1.0f, 1.0f, nrm, co); row[28 + j] = emit_scaled(gq[j], -1.0f, 1.0f, nrm, co); }
        if (ASYM) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { row[OD + j] = emit_scaled(cv[j], -0.5f, 0.5f, nrm, co); row[OD + 3 + j] = emit_scaled(cw[j], -0.5f, 0.5f, nrm, co); }
        }
        auto add_noise = [&]() {
            float nz[28];
#pragma unroll
            for (int b = 4; b < 7; ++b) rng4(P, gid, sa.frame, RNG_OBS_NOISE + (uint32_t)b, &nz[4 * b]);      // the blocks that hold slots 18..24
#pragma unroll
            for (int jj = 18; jj < 25; ++jj) row[jj] = f_clamp(FMA(P.dr_obs_noise, 2.0f * nz[jj] - 1.0f, row[jj]), -co, co);
        };
        if (!ASYM && P.dr_obs_noise > 0.0f) add_noise();
        STAMP(33);
        BAR();                                                  // P3: tile complete, fingertip exchange published
        STAMP(34);
        // ---- rewards (reference rewards.py; order of trifinger_env.py:513-550), termination (:1053-1099) ----
        if (!IS_RESET) {
            float tips[9], tip_prev[9];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { tips[3 * f + j] = LD(L_XCH + 6 * f + j); tip_prev[3 * f + j] = LD(L_XCH + 6 * f + 3 + j); }
            }
            const RewardCoef& rc = sa.rc;
            float r[6];
            {
                float s = 0.0f;
                s = s + (norm_p3(&tips[0], cp, P.norm_p) - norm_p3(&tip_prev[0], prev_obj, P.norm_p));
                s = s + (norm_p3(&tips[3], cp, P.norm_p) - norm_p3(&tip_prev[3], prev_obj, P.norm_p));
                s = s + (norm_p3(&tips[6], cp, P.norm_p) - norm_p3(&tip_prev[6], prev_obj, P.norm_p));
                r[0] = rc.c_reach * s;
            }
            {
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < 9; ++j) s = s + LD(L_VSQ + j);      // ((tips[j] - tip_prev[j]) / dt)^2, formed by the finger roles
                r[1] = rc.c_move_pen * s;
            }
            if (!WIDE) { o_dist = LD(L_OTERM); o_ang = LD(L_OTERM + 1); o_r[0] = LD(L_OTERM + 2); o_r[1] = LD(L_OTERM + 3); o_r[2] = LD(L_OTERM + 4); o_r[3] = LD(L_OTERM + 5); }
            const float dist = o_dist, ang = o_ang;
            r[2] = o_r[0]; r[3] = o_r[1]; r[4] = o_r[2]; r[5] = o_r[3];
            float total = 0.0f;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                r[t] = guarded ? 0.0f : r[t];
                if (P.rew_active[t]) { total = total + r[t]; st.rew[t] += cx.valid ? r[t] : 0.0f; }
            }
            bool pos_ok = dist <= P.pos_tol;
            bool ori_ok = ang <= P.ori_tol;
            st.pos_cnt += (cx.valid && pos_ok) ? 1.0f : 0.0f;
            st.ori_cnt += (cx.valid && ori_ok) ? 1.0f : 0.0f;
            bool done;
            if (P.task_difficulty < 4) done = pos_ok;
            else if (P.task_difficulty == 4) done = pos_ok && ori_ok;
            else done = ori_ok;
            bool succ = c_successes;
            if (P.success_activate) {
                if (done) total = total + P.success_bonus;
                if (cx.valid) P.goal_reset_buf[(unsigned)cx.i] = (uint8_t)done;
                c_goal_reset = done;
                succ = succ || done;
            } else {
                succ = c_goal_reset && succ;
            }
            c_successes = succ;
            if (cx.valid) {
                P.successes[(unsigned)cx.i] = (uint8_t)succ;
                P.reward[(unsigned)cx.i] = total;
            }
            st.succ += (cx.valid && succ) ? 1.0f : 0.0f;
        }
        STAMP(36);
        stats_begin(P, st, lane, tk);
        STAMP(37);
        if (ASYM && P.dr_obs_noise > 0.0f) {                    // (the tiles themselves leave through the three finger wavefronts: NT_F)
            BAR();                                              // P4
            add_noise();
            BAR();                                              // P5
        }
    }
    // ---- state rows of the cube role; the moving goal advances AFTER the step's observations and rewards used its pose
    // (trifinger_env.py:500-559: __update_goal_movement_post comes last) ----
    if (MODE & (M_RESETS | M_SIM | M_POST)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { STST(TF_S_CUBE_P + j, cp[j]); STST(TF_S_CUBE_V + j, cv[j]); STST(TF_S_CUBE_W + j, cw[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) STST(TF_S_CUBE_Q + j, cq[j]);
    }
    if (MODE & (M_RESETS | M_SIM)) {
#pragma unroll
        for (int j = 0; j < 12; ++j) STST(TF_S_LAM_CF + j, lam_cf[j]);
        STST(TF_S_CF_FACE, cf_face); STST(TF_S_CW_FACE, cw_face);
        if (__builtin_expect(cw_face != 0.0f, 0)) {
#pragma unroll
            for (int j = 0; j < 12; ++j) STST(TF_S_LAM_CW + j, lam_cw[j]);
        }
    }
    if ((MODE & M_POST) && P.goal_rotation_activate) {
        float gq2[4], gw2[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) gq2[j] = LDST(TF_S_GOAL_Q + j);
#pragma unroll
        for (int j = 0; j < 3; ++j) gw2[j] = LDST(TF_S_GOAL_W + j);
        const int nadv = ((MODE & M_SIM) ? sa.nsim : P.control_decimation) * P.substeps;
        for (int s = 0; s < nadv; ++s) quat_integrate(gq2, gw2, P.hsub);
#pragma unroll
        for (int j = 0; j < 4; ++j) STST(TF_S_GOAL_Q + j, gq2[j]);
    }
    STAMP(38);
    if (MODE & M_FINISH) {                                      // env_base.py:391-399
        if (cx.valid) {
            int sN = c_steps + 1;
            P.steps[(unsigned)cx.i] = (int64_t)sN;
            bool rb = c_reset;
            if (P.episode_length > 0 && sN >= P.episode_length) { rb = true; P.reset_buf[(unsigned)cx.i] = 1; }
            P.dones[(unsigned)cx.i] = (uint8_t)(rb && c_goal_reset);
        }
    }
    STAMP(39);
    if (MODE & M_POST) stats_end(P, lane, tk);
    STAMP(35);
    STAMPV(40, __builtin_amdgcn_s_getreg((31 << 11) | 4));      // HW_REG_HW_ID
    STAMPV(41, __builtin_amdgcn_s_getreg((31 << 11) | 20));     // HW_REG_XCC_ID
}
