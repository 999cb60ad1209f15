// trifinger_hip.hip - MI355X (gfx950) kernels + C ABI of the TriFinger vectorised environment step.
//
// The hot path of pairlab/leibnizgym (IsaacEnvBase.step -> TrifingerEnv hooks -> gymapi.simulate;
// reference leibnizgym/envs/env_base.py:345-401, leibnizgym/envs/trifinger/trifinger_env.py:373-559,959-1265)
// as ONE fused launch per control step:
//
//   masked reset / goal reset (Philox4x32-10 keyed by global env id)  ->  PD/torque law  ->
//   decimation x substeps x { 3 x 3-DoF articulated forward dynamics + free cube, contact generation,
//   projected Gauss-Seidel over contact + joint-limit rows, symplectic Euler }  ->
//   fingertip FK, obs[41]/states[113] assembly + normalisation, six reward terms, termination,
//   step counters / time-out / dones, wave-reduced episode statistics.
//
// Execution model (CDNA4): one environment per lane, one 64-lane wavefront per workgroup, state in HBM
// as structure-of-arrays rows [field][env] so that every global access of a wave is one coalesced 256-B
// line.  Per-env matrices are at most 3x3 / 6x6: no MFMA.  All contact rows of the solver (J, M^-1 J^T, directions,
// 1/D, impulses) live in the register file - the kernel is compiled for 1 wave per SIMD: at 65536 envs the chip
// holds exactly one wave per SIMD, so all 512 registers per lane are there to be used, and what binds is how fast
// one wave issues instructions (~5 cycles each), i.e. the instruction count.  LDS holds the
// [64][W] transposes of the row-major API tensors (action [N,A], obs [N,41], states [N,113]: global traffic stays
// coalesced, dwordx4) and, between them, the values that are cold during the solve.  Episode statistics are reduced
// with DPP butterflies per wave and folded across waves with fixed-point integer atomics (order independent, hence
// deterministic; the last arriving wave writes info[]): one launch per step, no host sync.
//
// Arithmetic contract (shared with the CPU oracle used by the tests): fp32 IEEE add/mul/div/sqrt, explicitly written
// fused multiply-adds and no compiler contraction (-ffp-contract=off), own polynomial sin/cos/exp/asin/log and Newton
// reciprocal / rsqrt, fixed evaluation order.  Per-env outputs are bit-identical to the oracle's.
//
// Physics is this build's own spec (the reference's lives in closed-source PhysX): DESIGN.md "Physics spec".
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/trifinger.h"

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
    gi32* steps;
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
    int32_t robot_reset_type, object_reset_type, goal_rotation_activate;
    float dof_pos_stddev, dof_vel_stddev, goal_rate;
    int32_t dr_enable;
    float dr_cube_mass[2], dr_cube_size[2], dr_friction[2], dr_motor[2], dr_link_mass[2], dr_restitution[2];
    float dr_obs_noise;      // half-width of the observation noise; 0 when off (or when dr_enable is 0)
    float dr_action_repeat;  // probability of re-applying the previous step's torque; 0 when off
    float clip_obs, clip_act; // fused wrapper clipping (tf_set_clipping); FLT_MAX when off
    int32_t rew_active[6];
    int32_t success_activate;
    float success_bonus, pos_tol, ori_tol;
    // stepping
    int32_t substeps, iters, control_decimation;
    float dt, hsub;
    float grav[3];
    TfModel m;
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
    uint32_t frame;          // frame count after this launch (counter of the observation-noise draws)
    uint32_t frame0;         // frame count at the start of the control step (counter of the action-repeat draw)
};


// ------------------------------------------------------------------------------------------------------
// deterministic elementary functions (Cephes single-precision polynomials; identical to the oracle's)
// ------------------------------------------------------------------------------------------------------
#define DEV __device__ __forceinline__

// Every workgroup of the env kernels is ONE wavefront.  LDS operations of a wave are executed in order by the LDS
// unit, so lane-to-lane hand-offs through LDS (tile transposes, contact rows) need no s_barrier and no counter
// drain - only the compiler must keep the program order.  A wavefront-scope fence does exactly that and emits no
// instruction; __syncthreads() would add `s_waitcnt vmcnt(0)` (waits for every outstanding global store) each time.
#define WAVE_LDS_ORDER() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

// Developer instrumentation (libtrifinger_hip_timing.so, used by tools/phase_timing.py only): lane 0 of every wave
// appends s_memtime stamps to scratch[wave*SCR_STRIDE + 16 ...]; the stamp counter lives in LDS.
#ifdef TF_PHASE_TIMING
#define SCR_STRIDE 80
__shared__ unsigned int g_stamp_ctr;
#define PHASE_STAMP_RESET() do { if (threadIdx.x == 0) g_stamp_ctr = 0; } while (0)
#define PHASE_STAMP() do { if (threadIdx.x == 0) { unsigned long long t_ = __builtin_readcyclecounter(); \
    unsigned int n_ = g_stamp_ctr; g_stamp_ctr = n_ + 1; if (n_ < SCR_STRIDE - 16) \
    ((gu32*)P.scratch)[(size_t)blockIdx.x * SCR_STRIDE + 16 + n_] = (unsigned int)t_; } } while (0)
#else
#define SCR_STRIDE 16
#define PHASE_STAMP_RESET() do { } while (0)
#define PHASE_STAMP() do { } while (0)
#endif

#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))

// one instruction each: v_min_f32 / v_max_f32 / v_med3_f32.  On non-NaN inputs they implement a total order with
// -0 < +0; the oracle's f_min / f_max / f_clamp restate exactly that (bitwise OR / AND of equal operands).
DEV float f_min(float a, float b) { return __builtin_fminf(a, b); }
DEV float f_max(float a, float b) { return __builtin_fmaxf(a, b); }
DEV float f_clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
DEV float f_abs(float a) { return __builtin_fabsf(a); }

DEV void tf_sincos(float x, float& s_out, float& c_out) {
    float k = __builtin_rintf(x * 0.63661977236758134f);
    int n = (int)k;
    float r = FMA(-k, 1.5703125f, x);
    r = FMA(-k, 4.837512969970703125e-4f, r);
    r = FMA(-k, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = FMA(FMA(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    ps = FMA(ps * z, r, r);
    float pc = FMA(FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    pc = FMA(pc * z, z, FMA(-0.5f, z, 1.0f));
    int q = n & 3;
    float s = (q & 1) ? pc : ps;
    float c = (q & 1) ? ps : pc;
    s_out = (q & 2) ? -s : s;
    c_out = (q == 1 || q == 2) ? -c : c;
}

DEV float tf_exp(float x) {
    x = f_clamp(x, -87.0f, 88.0f);
    float k = __builtin_rintf(x * 1.44269504088896341f);
    int n = (int)k;
    float r = FMA(-k, 0.693359375f, x);
    r = FMA(k, 2.12194440e-4f, r);
    float z = r * r;
    float p = FMA(FMA(FMA(FMA(FMA(1.9875691500e-4f, r, 1.3981999507e-3f), r, 8.3334519073e-3f), r, 4.1665795894e-2f), r,
                      1.6666665459e-1f), r, 5.0000001201e-1f);
    float e = FMA(p, z, r) + 1.0f;
    return e * __uint_as_float((uint32_t)(n + 127) << 23);
}

DEV float tf_asin(float x) {
    float a = f_abs(x);
    a = f_min(a, 1.0f);
    bool big = a > 0.5f;
    float z = big ? 0.5f * (1.0f - a) : a * a;
    float y = big ? __builtin_sqrtf(z) : a;
    float p = FMA(FMA(FMA(FMA(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                  1.6666752422e-1f);
    p = FMA(p * z, y, y);
    if (big) p = 1.5707963267948966f - (p + p);
    return (x < 0.0f) ? -p : p;
}

// Deterministic reciprocal / reciprocal square root for positive normal x, used for the physics-internal scalings
// (1/D of the contact rows, unit normals, 1/det ...): integer seed + 3 Newton steps in FMA arithmetic, ~1 ulp (rcp) and
// ~2 ulp (rsqrt).  Integer and fused multiply-add operations only, so both sides of the parity tests agree bit for
// bit, and the GPU issues neither the quarter-rate v_rcp/v_sqrt nor the IEEE division / square-root fix-up sequences
// (11 and 19 issue slots against 7 and 12).  Quantities that the reference defines (rewards, sampling) keep IEEE
// division and square root.
DEV float f_rcp(float x) {
    float r = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
    r = FMA(r, FMA(-x, r, 1.0f), r);
    r = FMA(r, FMA(-x, r, 1.0f), r);
    r = FMA(r, FMA(-x, r, 1.0f), r);
    return r;
}
// Two Newton steps (relative error ~2.4e-4) for the 1/D of a contact row: 1/D only scales the Gauss-Seidel update of that
// row, its fixed point (the complementarity solution) does not depend on it.
DEV float f_rcp2(float x) {
    float r = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
    r = FMA(r, FMA(-x, r, 1.0f), r);
    r = FMA(r, FMA(-x, r, 1.0f), r);
    return r;
}
DEV float f_rsqrt(float x) {
    float y = __uint_as_float(0x5F375A86u - (__float_as_uint(x) >> 1));
    const float h = 0.5f * x;
    y = y * FMA(-h, y * y, 1.5f);
    y = y * FMA(-h, y * y, 1.5f);
    y = y * FMA(-h, y * y, 1.5f);
    return y;
}

DEV float tf_log(float x) {
    uint32_t u = __float_as_uint(x);
    int e = (int)((u >> 23) & 0xff) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e = e - 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = FMA(FMA(FMA(FMA(FMA(FMA(FMA(FMA(7.0376836292e-2f, m, -1.1514610310e-1f), m, 1.1676998740e-1f), m,
                  -1.2420140846e-1f), m, 1.4249322787e-1f), m, -1.6668057665e-1f), m, 2.0000714765e-1f), m,
                  -2.4999993993e-1f), m, 3.3333331174e-1f);
    y = (y * m) * z;
    float fe = (float)e;
    y = FMA(fe, -2.12194440e-4f, y);
    y = FMA(-0.5f, z, y);
    float r = m + y;
    r = FMA(fe, 0.693359375f, r);
    return r;
}

DEV float f_sqrt(float x) { return __builtin_sqrtf(x); }
// Hide a value from the optimiser.  hipcc folds (0.0f - y) into -y, which turns +0 into -0 when y == +0
// (normalised action slot of a freshly reset env); an opaque operand keeps the IEEE subtraction.
DEV float opaque(float x) { asm volatile("" : "+v"(x)); return x; }

// ------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. SC'11) - counter = (global env id, reset count, stream tag, 0)
// ------------------------------------------------------------------------------------------------------
DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
DEV float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }
enum { RNG_OBJECT = 0, RNG_GOAL_POS = 1, RNG_GOAL_QUAT = 2, RNG_GOAL_ANGVEL = 3, RNG_ROBOT = 4, RNG_DR = 9 /* and 10 */,
       RNG_OBS_NOISE = 16 /* .. 22, counter = frame count instead of reset count */, RNG_ACT_REPEAT = 24 /* counter = frame count */ };
DEV void rng4(const DevParams& P, uint32_t gid, uint32_t count, uint32_t tag, float u[4]) {
    uint32_t r[4];
    philox4x32_10(gid, count, tag, 0u, P.seed_lo, P.seed_hi, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = u01(r[i]);
}
DEV void box_muller(float ua, float ub, float& n0, float& n1) {
    float r = f_sqrt(-2.0f * tf_log(1.0f - ua));
    float s, c;
    tf_sincos(6.2831855f * ub, s, c);
    n0 = r * c;
    n1 = r * s;
}

// ------------------------------------------------------------------------------------------------------
// small vector helpers
// ------------------------------------------------------------------------------------------------------
DEV void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = FMA(a[1], b[2], -(a[2] * b[1]));
    o[1] = FMA(a[2], b[0], -(a[0] * b[2]));
    o[2] = FMA(a[0], b[1], -(a[1] * b[0]));
}
DEV float dot3(const float a[3], const float b[3]) { return FMA(a[2], b[2], FMA(a[1], b[1], a[0] * b[0])); }
DEV void sym_mul(const float I[6], const float v[3], float o[3]) {   // xx yy zz xy xz yz
    o[0] = FMA(I[4], v[2], FMA(I[3], v[1], I[0] * v[0]));
    o[1] = FMA(I[5], v[2], FMA(I[1], v[1], I[3] * v[0]));
    o[2] = FMA(I[2], v[2], FMA(I[5], v[1], I[4] * v[0]));
}
DEV void sym3_mul(const float S[6], const float v[3], float o[3]) {  // 00 01 02 11 12 22
    o[0] = FMA(S[2], v[2], FMA(S[1], v[1], S[0] * v[0]));
    o[1] = FMA(S[4], v[2], FMA(S[3], v[1], S[1] * v[0]));
    o[2] = FMA(S[5], v[2], FMA(S[4], v[1], S[2] * v[0]));
}

// quaternions (xyzw): reference leibnizgym/utils/torch_utils.py:83-150
DEV void quat_mul(const float a[4], const float b[4], float o[4]) {
    float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    o[3] = qq - ww + (z1 - y1) * (y2 - z2);
    o[0] = qq - xx + (x1 + w1) * (x2 + w2);
    o[1] = qq - yy + (w1 - x1) * (y2 + z2);
    o[2] = qq - zz + (z1 + y1) * (w2 - x2);
}
DEV float quat_diff_rad(const float a[4], const float b[4]) {
    float bc[4] = {-b[0], -b[1], -b[2], b[3]};
    float m[4];
    quat_mul(a, bc, m);
    float nrm = f_sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    return 2.0f * tf_asin(f_min(nrm, 1.0f));
}
DEV float lgsk(float x, float scale) {      // reference rewards.py:20-34
    float s = x * scale;
    return 1.0f / (tf_exp(s) + 2.0f + tf_exp(-s));
}
DEV void quat_to_rot(const float q[4], float R[9]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = FMA(-2.0f, FMA(y, y, z * z), 1.0f); R[1] = 2.0f * FMA(x, y, -(w * z));   R[2] = 2.0f * FMA(x, z, w * y);
    R[3] = 2.0f * FMA(x, y, w * z);            R[4] = FMA(-2.0f, FMA(x, x, z * z), 1.0f); R[5] = 2.0f * FMA(y, z, -(w * x));
    R[6] = 2.0f * FMA(x, z, -(w * y));         R[7] = 2.0f * FMA(y, z, w * x);      R[8] = FMA(-2.0f, FMA(x, x, y * y), 1.0f);
}
DEV void quat_integrate(float q[4], const float w[3], float h) {
    float hx = 0.5f * h * w[0], hy = 0.5f * h * w[1], hz = 0.5f * h * w[2];
    float x = q[0], y = q[1], z = q[2], s = q[3];
    float nx = x + FMA(hx, s, FMA(hy, z, -(hz * y)));
    float ny = y + FMA(hy, s, FMA(hz, x, -(hx * z)));
    float nz = z + FMA(hz, s, FMA(hx, y, -(hy * x)));
    float ns = s - FMA(hx, x, FMA(hy, y, hz * z));
    float inv = f_rsqrt(FMA(nx, nx, FMA(ny, ny, FMA(nz, nz, ns * ns))));
    q[0] = nx * inv; q[1] = ny * inv; q[2] = nz * inv; q[3] = ns * inv;
}

// ------------------------------------------------------------------------------------------------------
// finger kinematics / dynamics in the finger base frame (world = Rz(yaw) base + (0,0,H))
// ------------------------------------------------------------------------------------------------------
struct FK {
    float s1, c1, s2, c2, s23, c23;
    float p2[3], p3[3];
    float ax[3];
    float Minv[6];
};

template <int LINK> DEV void rot_link(const FK& k, const float u[3], float o[3]) {
    float wx = u[0], wy = u[1], wz = u[2];
    if (LINK >= 2) {
        float ca = (LINK == 2) ? k.c2 : k.c23, sa = (LINK == 2) ? k.s2 : k.s23;
        float ty = FMA(ca, u[1], -(sa * u[2]));
        float tz = FMA(sa, u[1], ca * u[2]);
        wy = ty; wz = tz;
    }
    o[0] = FMA(k.c1, wx, k.s1 * wz);
    o[1] = wy;
    o[2] = FMA(k.c1, wz, -(k.s1 * wx));
}
template <int LINK> DEV void rot_link_T(const FK& k, const float v[3], float o[3]) {
    float wx = FMA(k.c1, v[0], -(k.s1 * v[2]));
    float wy = v[1];
    float wz = FMA(k.s1, v[0], k.c1 * v[2]);
    if (LINK >= 2) {
        float ca = (LINK == 2) ? k.c2 : k.c23, sa = (LINK == 2) ? k.s2 : k.s23;
        float ty = FMA(ca, wy, sa * wz);
        float tz = FMA(ca, wz, -(sa * wy));
        wy = ty; wz = tz;
    }
    o[0] = wx; o[1] = wy; o[2] = wz;
}

DEV void fk_setup(const TfModel& m, const float q[3], FK& k) {
    tf_sincos(q[0], k.s1, k.c1);
    tf_sincos(q[1], k.s2, k.c2);
    tf_sincos(q[1] + q[2], k.s23, k.c23);
    k.ax[0] = k.c1; k.ax[1] = 0.0f; k.ax[2] = -k.s1;
    rot_link<1>(k, m.j2_origin, k.p2);
    float t[3];
    rot_link<2>(k, m.j3_origin, t);
    k.p3[0] = k.p2[0] + t[0]; k.p3[1] = k.p2[1] + t[1]; k.p3[2] = k.p2[2] + t[2];
}

DEV void levers(const FK& k, const float P[3], float L1[3], float L2[3], float L3[3]) {
    L1[0] = P[2]; L1[1] = 0.0f; L1[2] = -P[0];
    float r2[3] = {P[0] - k.p2[0], P[1] - k.p2[1], P[2] - k.p2[2]};
    float r3[3] = {P[0] - k.p3[0], P[1] - k.p3[1], P[2] - k.p3[2]};
    cross3(k.ax, r2, L2);
    cross3(k.ax, r3, L3);
}

// Joint-space mass matrix M (00 01 02 11 12 22) and bias h = C(q,qd) qd + g(q); grav = gravity vector (base frame).
// Evaluated in the coordinates of link 1 ("frame A": the base frame turned by joint 1 about y).  There joint 1 is the
// y axis, joints 2 and 3 are the x axis, links 2 and 3 turn about x by q2 and q2+q3, and every vector of the recursive
// Newton-Euler pass has structural zeros: w_k = (a_k, w, 0) with a_2 = qd2, a_3 = qd2 + qd3 and dw_k = (0, 0, -w a_k),
// hence  dw x r + w x (w x r) = (w (2 a r_y - w r_x), -a^2 r_y, -(a^2 + w^2) r_z).  Only the components that reach
// the three joint torques (n1_y, n2_x, n3_x) are formed.  tests/test_physics_analytic.py checks M against the fp64
// kinetic energy and h against the Lagrangian derivatives of an independent model.
DEV void finger_dynamics(const TfModel& m, const FK& k, const float qd[3], const float grav[3], float M[6], float bias[3]) {
    const float m1 = m.link_mass[0], m2 = m.link_mass[1], m3 = m.link_mass[2];
    const float* I1 = m.link_inertia[0];
    const float* I2 = m.link_inertia[1];
    const float* I3 = m.link_inertia[2];
    const float* p2 = m.j2_origin;               /* joint-2 origin and link-1 COM are constants of frame A */
    const float* c1 = m.link_com[0];
    /* frame-A geometry: Rx(a) v = (v_x, c v_y - s v_z, s v_y + c v_z) */
    float d23[3], b[3], e3[3], e2[3];
    d23[0] = m.j3_origin[0];                     /* joint 2 -> joint 3 */
    d23[1] = FMA(k.c2, m.j3_origin[1], -(k.s2 * m.j3_origin[2]));
    d23[2] = FMA(k.s2, m.j3_origin[1], k.c2 * m.j3_origin[2]);
    b[0] = m.link_com[1][0];                     /* joint 2 -> COM 2 */
    b[1] = FMA(k.c2, m.link_com[1][1], -(k.s2 * m.link_com[1][2]));
    b[2] = FMA(k.s2, m.link_com[1][1], k.c2 * m.link_com[1][2]);
    e3[0] = m.link_com[2][0];                    /* joint 3 -> COM 3 */
    e3[1] = FMA(k.c23, m.link_com[2][1], -(k.s23 * m.link_com[2][2]));
    e3[2] = FMA(k.s23, m.link_com[2][1], k.c23 * m.link_com[2][2]);
    e2[0] = d23[0] + e3[0]; e2[1] = d23[1] + e3[1]; e2[2] = d23[2] + e3[2];     /* joint 2 -> COM 3 */
    const float c2x = p2[0] + b[0], c2z = p2[2] + b[2];                           /* COM 2 (x, z) */
    const float c3x = p2[0] + e2[0], c3z = p2[2] + e2[2];                         /* COM 3 (x, z) */
    /* ---- mass matrix: linear part from the COM lever arms L1 = y x P = (P_z, 0, -P_x), L2/L3 = x x r = (0, -r_z, r_y);
     * angular part from the joint axes seen in the link frames, y -> (0, c, -s), x -> x ---- */
    const float u2I = FMA(k.s2 * k.s2, I2[2], FMA(k.c2 * k.c2, I2[1], ((-2.0f * k.c2) * k.s2) * I2[5]));
    const float u3I = FMA(k.s23 * k.s23, I3[2], FMA(k.c23 * k.c23, I3[1], ((-2.0f * k.c23) * k.s23) * I3[5]));
    const float u2x = FMA(k.c2, I2[3], -(k.s2 * I2[4]));
    const float u3x = FMA(k.c23, I3[3], -(k.s23 * I3[4]));
    M[0] = FMA(m3, FMA(c3x, c3x, c3z * c3z), FMA(m2, FMA(c2x, c2x, c2z * c2z), m1 * FMA(c1[0], c1[0], c1[2] * c1[2])))
           + ((I1[1] + u2I) + u3I);
    M[1] = (u2x + u3x) - FMA(m3 * c3x, e2[1], (m2 * c2x) * b[1]);
    M[2] = FMA(-(m3 * c3x), e3[1], u3x);
    M[3] = FMA(m3, FMA(e2[1], e2[1], e2[2] * e2[2]), FMA(m2, FMA(b[1], b[1], b[2] * b[2]), I2[0] + I3[0]));
    M[4] = FMA(m3, FMA(e2[1], e3[1], e2[2] * e3[2]), I3[0]);
    M[5] = FMA(m3, FMA(e3[1], e3[1], e3[2] * e3[2]), I3[0]);
    /* ---- recursive Newton-Euler with zero joint acceleration, base acceleration = -gravity (in frame A) ---- */
    const float w = qd[0], a2 = qd[1], a3 = qd[1] + qd[2];
    const float ww = w * w;
    float a0[3];
    a0[0] = FMA(k.s1, grav[2], -(k.c1 * grav[0]));
    a0[1] = -grav[1];
    a0[2] = -FMA(k.s1, grav[0], k.c1 * grav[2]);
    /* link 1 (a = 0): COM force (x, z only: F1_y never reaches a joint torque), acceleration of joint 2 */
    const float F1x = m1 * FMA(-ww, c1[0], a0[0]);
    const float F1z = m1 * FMA(-ww, c1[2], a0[2]);
    float A2[3] = {FMA(-ww, p2[0], a0[0]), a0[1], FMA(-ww, p2[2], a0[2])};
    /* link 2: offset(r) = (w (2 a r_y - w r_x), -a^2 r_y, -(a^2 + w^2) r_z) */
    const float aa2 = a2 * a2, sw2 = aa2 + ww, ta2 = a2 + a2;
    float A3[3], F2[3], F3[3];
    A3[0] = FMA(w, FMA(ta2, d23[1], -(w * d23[0])), A2[0]);
    A3[1] = FMA(-aa2, d23[1], A2[1]);
    A3[2] = FMA(-sw2, d23[2], A2[2]);
    F2[0] = m2 * FMA(w, FMA(ta2, b[1], -(w * b[0])), A2[0]);
    F2[1] = m2 * FMA(-aa2, b[1], A2[1]);
    F2[2] = m2 * FMA(-sw2, b[2], A2[2]);
    /* link 3 */
    const float aa3 = a3 * a3, sw3 = aa3 + ww, ta3 = a3 + a3;
    F3[0] = m3 * FMA(w, FMA(ta3, e3[1], -(w * e3[0])), A3[0]);
    F3[1] = m3 * FMA(-aa3, e3[1], A3[1]);
    F3[2] = m3 * FMA(-sw3, e3[2], A3[2]);
    /* inertial moments N = I dw + w x I w in the link frames (w_l = (a, c w, -s w), dw_l = (0, s d, c d), d = -w a),
     * turned back to frame A; only x and y are needed */
    float N2x, N2y, N3x, N3y;
    {
        const float d = -(w * a2);
        float wl[3] = {a2, k.c2 * w, -(k.s2 * w)}, dl1 = k.s2 * d, dl2 = k.c2 * d;
        float Iw[3], Id[3], t[3];
        sym_mul(I2, wl, Iw);
        Id[0] = FMA(I2[4], dl2, I2[3] * dl1);
        Id[1] = FMA(I2[5], dl2, I2[1] * dl1);
        Id[2] = FMA(I2[2], dl2, I2[5] * dl1);
        cross3(wl, Iw, t);
        const float n0 = Id[0] + t[0], n1 = Id[1] + t[1], n2 = Id[2] + t[2];
        N2x = n0;
        N2y = FMA(k.c2, n1, -(k.s2 * n2));
    }
    {
        const float d = -(w * a3);
        float wl[3] = {a3, k.c23 * w, -(k.s23 * w)}, dl1 = k.s23 * d, dl2 = k.c23 * d;
        float Iw[3], Id[3], t[3];
        sym_mul(I3, wl, Iw);
        Id[0] = FMA(I3[4], dl2, I3[3] * dl1);
        Id[1] = FMA(I3[5], dl2, I3[1] * dl1);
        Id[2] = FMA(I3[2], dl2, I3[5] * dl1);
        cross3(wl, Iw, t);
        const float n0 = Id[0] + t[0], n1 = Id[1] + t[1], n2 = Id[2] + t[2];
        N3x = n0;
        N3y = FMA(k.c23, n1, -(k.s23 * n2));
    }
    /* backward pass, moments about the joint origins: x and y components only */
    const float n3x = N3x + FMA(e3[1], F3[2], -(e3[2] * F3[1]));
    const float n3y = N3y + FMA(e3[2], F3[0], -(e3[0] * F3[2]));
    const float n2x = ((N2x + FMA(b[1], F2[2], -(b[2] * F2[1]))) + n3x) + FMA(d23[1], F3[2], -(d23[2] * F3[1]));
    const float n2y = ((N2y + FMA(b[2], F2[0], -(b[0] * F2[2]))) + n3y) + FMA(d23[2], F3[0], -(d23[0] * F3[2]));
    const float f2x = F2[0] + F3[0], f2z = F2[2] + F3[2];
    const float n1y = (FMA(c1[2], F1x, -(c1[0] * F1z)) + n2y) + FMA(p2[2], f2x, -(p2[0] * f2z));
    bias[0] = n1y;
    bias[1] = n2x;
    bias[2] = n3x;
}

DEV void inv3sym(const float M[6], float Mi[6]) {
    float A = FMA(M[3], M[5], -(M[4] * M[4]));
    float B = FMA(M[2], M[4], -(M[1] * M[5]));
    float C = FMA(M[1], M[4], -(M[2] * M[3]));
    float det = FMA(M[2], C, FMA(M[1], B, M[0] * A));
    float rd = f_rcp(det);
    Mi[0] = A * rd; Mi[1] = B * rd; Mi[2] = C * rd;
    Mi[3] = FMA(M[0], M[5], -(M[2] * M[2])) * rd;
    Mi[4] = FMA(M[1], M[2], -(M[0] * M[4])) * rd;
    Mi[5] = FMA(M[0], M[3], -(M[1] * M[1])) * rd;
}

template <int F> DEV void base_to_world(const TfModel& m, const float b[3], float w[3]) {
    float c = m.base_yaw_cos[F], s = m.base_yaw_sin[F];
    w[0] = FMA(c, b[0], -(s * b[1]));
    w[1] = FMA(s, b[0], c * b[1]);
    w[2] = b[2] + m.base_height;
}
template <int F> DEV void dir_world_to_base(const TfModel& m, const float w[3], float b[3]) {
    float c = m.base_yaw_cos[F], s = m.base_yaw_sin[F];
    b[0] = FMA(c, w[0], s * w[1]);
    b[1] = FMA(c, w[1], -(s * w[0]));
    b[2] = w[2];
}
template <int F> DEV void dir_base_to_world(const TfModel& m, const float b[3], float w[3]) {
    float c = m.base_yaw_cos[F], s = m.base_yaw_sin[F];
    w[0] = FMA(c, b[0], -(s * b[1]));
    w[1] = FMA(s, b[0], c * b[1]);
    w[2] = b[2];
}

// o = R v and o = R^T v for a row-major 3x3
DEV void mat3_mul(const float R[9], const float v[3], float o[3]) {
    o[0] = FMA(R[2], v[2], FMA(R[1], v[1], R[0] * v[0]));
    o[1] = FMA(R[5], v[2], FMA(R[4], v[1], R[3] * v[0]));
    o[2] = FMA(R[8], v[2], FMA(R[7], v[1], R[6] * v[0]));
}
DEV void mat3T_mul(const float R[9], const float v[3], float o[3]) {
    o[0] = FMA(R[6], v[2], FMA(R[3], v[1], R[0] * v[0]));
    o[1] = FMA(R[7], v[2], FMA(R[4], v[1], R[1] * v[0]));
    o[2] = FMA(R[8], v[2], FMA(R[5], v[1], R[2] * v[0]));
}

DEV void tangent_basis(const float n[3], float t1[3], float t2[3]) {
    if (f_abs(n[2]) < 0.9f) {
        float inv = f_rsqrt(FMA(n[0], n[0], n[1] * n[1]));
        t1[0] = -n[1] * inv; t1[1] = n[0] * inv; t1[2] = 0.0f;
    } else {
        float inv = f_rsqrt(FMA(n[1], n[1], n[2] * n[2]));
        t1[0] = 0.0f; t1[1] = -n[2] * inv; t1[2] = n[1] * inv;
    }
    cross3(n, t1, t2);
}

DEV float contact_bias(const TfModel& m, float gap, float vn0, float inv_h, float restitution) {
    float b;
    if (gap >= 0.0f) b = gap * inv_h;
    else b = f_max(m.erp * gap * inv_h, -m.max_depenetration_velocity);
    if (restitution > 0.0f && gap < m.contact_offset && vn0 < -m.bounce_threshold) b = f_min(b, restitution * vn0);
    return b;
}

// ------------------------------------------------------------------------------------------------------
// per-env working state (registers)
// ------------------------------------------------------------------------------------------------------
struct Env {
    float q[9], qd[9];
    float cp[3], cq[4], cv[3], cw[3];
    float gp[3], gq[4], gw[3];
    float tau[9];
    float ft[18];
    float dr[TF_NUM_DR];   // domain-randomisation scale factors: cube mass, cube size, friction, motor torque, link mass, restitution
};

// LDS is used for the row-major API tiles only ([64][W] transposes); W <= MAX_STATES.
#define LDS_FLOATS (WAVE * MAX_STATES)

// One finger contact (capsule-cube or tip-floor): three rows (normal + two tangents).  The Jacobian rows J, M^-1 J^T,
// the world directions and the cube arm are loop-invariant over the solver sweeps and live in the register file
// (the allocator parks the cold part in AGPRs: v_accvgpr_read, no wait counters) - measured faster than re-reading
// them from LDS every sweep with only one wave per SIMD to hide the LDS round trip.
struct FingerContactRegs {
    bool active;
    float Jf[9], Wf[9];      // row d at [3d .. 3d+2]
    float dir[9];            // world n, t1, t2 (capsule-cube contact only)
    float rc[3];             // cube arm (capsule-cube contact only)
    float Dinv[3];
    float bias;
    float arm[3];
    float lam[3];
};
// one cube corner against the floor or the wall.  Rows are always evaluated (no branch): an inactive contact has
// Dinv = 0 and bias = 0, so its impulses stay exactly zero.
struct CubeContactRegs {
    float r[3];
    float n[2];          // wall contacts: horizontal inward normal
    float Dinv[3];
    float bias;
    float lam[3];
};

// rows of one finger contact: point Pb (base frame), world normal, cube arm rc
template <int F, bool WITH_CUBE>
DEV void finger_rows(const TfModel& m, const FK& k, const float Pb[3], const float n_w[3], const float rc[3],
                     float inv_m, float inv_I, FingerContactRegs& c) {
    float t1[3], t2[3];
    tangent_basis(n_w, t1, t2);
    float L1[3], L2[3], L3[3];
    levers(k, Pb, L1, L2, L3);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float* dw = (d == 0) ? n_w : ((d == 1) ? t1 : t2);
        float db[3];
        float* Jf = &c.Jf[3 * d];
        float* Wf = &c.Wf[3 * d];
        dir_world_to_base<F>(m, dw, db);
        Jf[0] = dot3(L1, db); Jf[1] = dot3(L2, db); Jf[2] = dot3(L3, db);
        sym3_mul(k.Minv, Jf, Wf);
        float D = dot3(Jf, Wf);
        if (WITH_CUBE) {
            float rxd[3];
            cross3(rc, dw, rxd);
            D = FMA(dot3(rxd, rxd), inv_I, D + inv_m);
            c.dir[3 * d] = dw[0]; c.dir[3 * d + 1] = dw[1]; c.dir[3 * d + 2] = dw[2];
        }
        c.Dinv[d] = f_rcp2(D);
    }
    if (WITH_CUBE) { c.rc[0] = rc[0]; c.rc[1] = rc[1]; c.rc[2] = rc[2]; }
}

DEV void cube_corner(const float R[9], float hc, int k, float sk, int idx, float r[3]) {
    // axes a < b are the two that are not k
    float y[3];
    float sa = (idx & 1) ? hc : -hc;
    float sb = (idx & 2) ? hc : -hc;
    float fk_ = sk * hc;
    y[0] = (k == 0) ? fk_ : sa;
    y[1] = (k == 1) ? fk_ : ((k == 0) ? sa : sb);
    y[2] = (k == 2) ? fk_ : sb;
    mat3_mul(R, y, r);
}

// ---- PGS row kernels (identical arithmetic in the oracle) ----
DEV float solve_normal(float& lam, float Dinv, float vrel, float bias) {
    float ln = f_max(FMA(-Dinv, vrel + bias, lam), 0.0f);
    float dl = ln - lam;
    lam = ln;
    return dl;
}
DEV float solve_tangent(float& lam, float Dinv, float vrel, float lim) {
    float ln = f_clamp(FMA(-Dinv, vrel, lam), -lim, lim);
    float dl = ln - lam;
    lam = ln;
    return dl;
}
// axis-aligned rows of a cube corner with arm r: direction +z / +x / +y
template <int SLOT, bool IS_NORMAL>
DEV void cube_row_z(CubeContactRegs& c, float mu, float inv_m, float inv_I, float v[3], float w[3]) {
    const float* r = c.r;
    float vrel = FMA(r[1], w[0], FMA(-r[0], w[1], v[2]));
    float dl = IS_NORMAL ? solve_normal(c.lam[SLOT], c.Dinv[SLOT], vrel, c.bias)
                         : solve_tangent(c.lam[SLOT], c.Dinv[SLOT], vrel, mu * c.lam[0]);
    float s = dl * inv_m, q = dl * inv_I;
    v[2] = v[2] + s;
    w[0] = FMA(r[1], q, w[0]);
    w[1] = FMA(-r[0], q, w[1]);
}
template <int SLOT>
DEV void cube_row_x(CubeContactRegs& c, float mu, float inv_m, float inv_I, float v[3], float w[3]) {
    const float* r = c.r;
    float vrel = FMA(r[2], w[1], FMA(-r[1], w[2], v[0]));
    float dl = solve_tangent(c.lam[SLOT], c.Dinv[SLOT], vrel, mu * c.lam[0]);
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = v[0] + s;
    w[1] = FMA(r[2], q, w[1]);
    w[2] = FMA(-r[1], q, w[2]);
}
template <int SLOT>
DEV void cube_row_y(CubeContactRegs& c, float mu, float inv_m, float inv_I, float v[3], float w[3]) {
    const float* r = c.r;
    float vrel = FMA(-r[2], w[0], FMA(r[0], w[2], v[1]));
    float dl = solve_tangent(c.lam[SLOT], c.Dinv[SLOT], vrel, mu * c.lam[0]);
    float s = dl * inv_m, q = dl * inv_I;
    v[1] = v[1] + s;
    w[0] = FMA(-r[2], q, w[0]);
    w[2] = FMA(r[0], q, w[2]);
}
// wall rows: inward horizontal normal n = (n0, n1, 0) and tangent t = (-n1, n0, 0)
DEV void wall_arm_n(const CubeContactRegs& c, float a[3]) {
    const float* r = c.r;
    a[0] = -(r[2] * c.n[1]);
    a[1] = r[2] * c.n[0];
    a[2] = FMA(r[0], c.n[1], -(r[1] * c.n[0]));
}
DEV void wall_arm_t(const CubeContactRegs& c, float b[3]) {
    const float* r = c.r;
    b[0] = -(r[2] * c.n[0]);
    b[1] = -(r[2] * c.n[1]);
    b[2] = FMA(r[0], c.n[0], r[1] * c.n[1]);
}
DEV void wall_row_n(CubeContactRegs& c, float inv_m, float inv_I, float v[3], float w[3]) {
    float a[3];
    wall_arm_n(c, a);
    float vrel = FMA(a[2], w[2], FMA(a[1], w[1], FMA(a[0], w[0], FMA(c.n[1], v[1], c.n[0] * v[0]))));
    float dl = solve_normal(c.lam[0], c.Dinv[0], vrel, c.bias);
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = FMA(c.n[0], s, v[0]);
    v[1] = FMA(c.n[1], s, v[1]);
    w[0] = FMA(a[0], q, w[0]); w[1] = FMA(a[1], q, w[1]); w[2] = FMA(a[2], q, w[2]);
}
DEV void wall_row_t(CubeContactRegs& c, float mu, float inv_m, float inv_I, float v[3], float w[3]) {
    float b[3];
    wall_arm_t(c, b);
    float vrel = FMA(b[2], w[2], FMA(b[1], w[1], FMA(b[0], w[0], FMA(c.n[0], v[1], -(c.n[1] * v[0])))));
    float dl = solve_tangent(c.lam[1], c.Dinv[1], vrel, mu * c.lam[0]);
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = FMA(-c.n[1], s, v[0]);
    v[1] = FMA(c.n[0], s, v[1]);
    w[0] = FMA(b[0], q, w[0]); w[1] = FMA(b[1], q, w[1]); w[2] = FMA(b[2], q, w[2]);
}

DEV void finger_contact_zero(FingerContactRegs& c) {
    c.active = false;
#pragma unroll
    for (int j = 0; j < 9; ++j) { c.Jf[j] = 0.0f; c.Wf[j] = 0.0f; c.dir[j] = 0.0f; }
    c.rc[0] = 0.0f; c.rc[1] = 0.0f; c.rc[2] = 0.0f;
    c.lam[0] = 0.0f; c.lam[1] = 0.0f; c.lam[2] = 0.0f;
    c.bias = 0.0f;
    c.Dinv[0] = 0.0f; c.Dinv[1] = 0.0f; c.Dinv[2] = 0.0f;
    c.arm[0] = 0.0f; c.arm[1] = 0.0f; c.arm[2] = 0.0f;
}
DEV void cube_contact_zero(CubeContactRegs& c) {
    c.lam[0] = 0.0f; c.lam[1] = 0.0f; c.lam[2] = 0.0f;
    c.bias = 0.0f; c.n[0] = 0.0f; c.n[1] = 0.0f;
    c.Dinv[0] = 0.0f; c.Dinv[1] = 0.0f; c.Dinv[2] = 0.0f;
}

// ---- contact generation for finger F (capsule vs cube, tip vs floor) ----
template <int F>
DEV void finger_contacts(const DevParams& P, const Env& e, const FK& k, const float R[9], const float* vq,
                         const float v[3], const float w[3], float hc, float inv_h, float inv_m, float inv_I,
                         FingerContactRegs& c, FingerContactRegs& g) {
    const TfModel& m = P.m;
    float t[3], Ab[3], Bb[3], Aw[3], Bw[3], To[3], Tw[3];
    rot_link<3>(k, m.cap_a, t);
    Ab[0] = k.p3[0] + t[0]; Ab[1] = k.p3[1] + t[1]; Ab[2] = k.p3[2] + t[2];
    rot_link<3>(k, m.cap_b, t);
    Bb[0] = k.p3[0] + t[0]; Bb[1] = k.p3[1] + t[1]; Bb[2] = k.p3[2] + t[2];
    rot_link<3>(k, m.tip_origin, t);
    To[0] = k.p3[0] + t[0]; To[1] = k.p3[1] + t[1]; To[2] = k.p3[2] + t[2];
    base_to_world<F>(m, Ab, Aw);
    base_to_world<F>(m, Bb, Bw);
    base_to_world<F>(m, To, Tw);
    // capsule (distal link) vs cube: closest points by alternating projection in the cube frame
    finger_contact_zero(c);
    float da[3] = {Aw[0] - e.cp[0], Aw[1] - e.cp[1], Aw[2] - e.cp[2]};
    float db[3] = {Bw[0] - e.cp[0], Bw[1] - e.cp[1], Bw[2] - e.cp[2]};
    float a[3], b[3];
    mat3T_mul(R, da, a);
    mat3T_mul(R, db, b);
    float d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    float inv_dd = f_rcp(dot3(d, d));
    float s = 1.0f, x[3], y[3];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { x[i] = FMA(s, d[i], a[i]); y[i] = f_clamp(x[i], -hc, hc); }
        float ya[3] = {y[0] - a[0], y[1] - a[1], y[2] - a[2]};
        s = f_clamp(dot3(ya, d) * inv_dd, 0.0f, 1.0f);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { x[i] = FMA(s, d[i], a[i]); y[i] = f_clamp(x[i], -hc, hc); }
    float ev[3] = {x[0] - y[0], x[1] - y[1], x[2] - y[2]};
    float dist2 = dot3(ev, ev);
    float nc[3], gap;
    if (__builtin_expect(dist2 > 1e-12f, 1)) {
        float inv = f_rsqrt(dist2);
        float dist = dist2 * inv;
        nc[0] = ev[0] * inv; nc[1] = ev[1] * inv; nc[2] = ev[2] * inv;
        gap = dist - m.cap_radius;
    } else {
        int bi = 0;
        float best = f_abs(x[0]) - hc;
        float p1 = f_abs(x[1]) - hc;
        if (p1 > best) { best = p1; bi = 1; }
        float p2 = f_abs(x[2]) - hc;
        if (p2 > best) { best = p2; bi = 2; }
        float xb = (bi == 0) ? x[0] : ((bi == 1) ? x[1] : x[2]);
        float sg = (xb < 0.0f) ? -1.0f : 1.0f;
        nc[0] = (bi == 0) ? sg : 0.0f; nc[1] = (bi == 1) ? sg : 0.0f; nc[2] = (bi == 2) ? sg : 0.0f;
        y[0] = (bi == 0) ? sg * hc : y[0]; y[1] = (bi == 1) ? sg * hc : y[1]; y[2] = (bi == 2) ? sg * hc : y[2];
        gap = best - m.cap_radius;
    }
    if (__builtin_expect(gap < m.contact_margin, 1)) {
        float n_w[3], rc[3], xw[3];
        mat3_mul(R, nc, n_w);
        mat3_mul(R, y, rc);
        mat3_mul(R, x, xw);
        float Pw[3] = {FMA(-m.cap_radius, n_w[0], e.cp[0] + xw[0]), FMA(-m.cap_radius, n_w[1], e.cp[1] + xw[1]),
                       FMA(-m.cap_radius, n_w[2], e.cp[2] + xw[2])};
        float Pr[3] = {Pw[0], Pw[1], Pw[2] - m.base_height};
        float Pb[3];
        dir_world_to_base<F>(m, Pr, Pb);
        c.active = true;
        finger_rows<F, true>(m, k, Pb, n_w, rc, inv_m, inv_I, c);
#pragma unroll
        for (int i = 0; i < 3; ++i) c.arm[i] = Pw[i] - Tw[i];
        float rxn[3];
        cross3(rc, &c.dir[0], rxn);
        float vn0 = dot3(&c.Jf[0], &vq[3 * F]) - (dot3(&c.dir[0], v) + dot3(rxn, w));
        c.bias = contact_bias(m, gap, vn0, inv_h, m.restitution_finger * e.dr[5]);
    }
    // tip sphere vs floor
    finger_contact_zero(g);
    float gapf = Bw[2] - m.cap_radius;
    if (__builtin_expect(gapf < m.contact_margin, 1)) {
        float n_w[3] = {0.0f, 0.0f, 1.0f}, zero[3] = {0.0f, 0.0f, 0.0f};
        float Pb[3] = {Bb[0], Bb[1], Bb[2] - m.cap_radius};
        float Pw[3] = {Bw[0], Bw[1], Bw[2] - m.cap_radius};
        g.active = true;
        finger_rows<F, false>(m, k, Pb, n_w, zero, inv_m, inv_I, g);
#pragma unroll
        for (int i = 0; i < 3; ++i) g.arm[i] = Pw[i] - Tw[i];
        float vn0 = dot3(&g.Jf[0], &vq[3 * F]);
        g.bias = contact_bias(m, gapf, vn0, inv_h, m.restitution_finger * e.dr[5]);
    }
}

// PGS rows of the finger-cube contact of finger F
template <int F>
DEV void solve_finger_cube(float mu, FingerContactRegs& c, float* vq, float v[3], float w[3], float inv_m, float inv_I) {
    if (__builtin_expect(!c.active, 0)) return;      // likely path falls through: no taken branch, no fetch bubble
    float* vf = &vq[3 * F];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float* Jf = &c.Jf[3 * d];
        const float* Wf = &c.Wf[3 * d];
        const float* dir = &c.dir[3 * d];
        float rxd[3];
        cross3(c.rc, dir, rxd);
        float vrel = dot3(Jf, vf) - (dot3(dir, v) + dot3(rxd, w));
        float dl = (d == 0) ? solve_normal(c.lam[0], c.Dinv[0], vrel, c.bias)
                            : solve_tangent(c.lam[d], c.Dinv[d], vrel, mu * c.lam[0]);
#pragma unroll
        for (int j = 0; j < 3; ++j) vf[j] = FMA(Wf[j], dl, vf[j]);
        float sc = dl * inv_m, q = dl * inv_I;
#pragma unroll
        for (int j = 0; j < 3; ++j) { v[j] = FMA(-dir[j], sc, v[j]); w[j] = FMA(-rxd[j], q, w[j]); }
    }
}
template <int F>
DEV void solve_tip_floor(float mu, FingerContactRegs& c, float* vq) {
    if (__builtin_expect(!c.active, 0)) return;      // likely path falls through: no taken branch, no fetch bubble
    float* vf = &vq[3 * F];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float* Jf = &c.Jf[3 * d];
        const float* Wf = &c.Wf[3 * d];
        float vrel = dot3(Jf, vf);
        float dl = (d == 0) ? solve_normal(c.lam[0], c.Dinv[0], vrel, c.bias)
                            : solve_tangent(c.lam[d], c.Dinv[d], vrel, mu * c.lam[0]);
#pragma unroll
        for (int j = 0; j < 3; ++j) vf[j] = FMA(Wf[j], dl, vf[j]);
    }
}

// wrench of one finger contact, world frame, about the tip-link origin
template <bool WITH_CUBE>
DEV void add_wrench(const FingerContactRegs& c, float inv_h, float* ft) {
    if (__builtin_expect(!c.active, 0)) return;      // likely path falls through: no taken branch, no fetch bubble
    float F[3];
    if (WITH_CUBE) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            F[i] = FMA(c.dir[6 + i], c.lam[2], FMA(c.dir[3 + i], c.lam[1], c.dir[i] * c.lam[0])) * inv_h;
    } else {
        // floor contact directions are constants: n = +z, t1 = -y, t2 = +x (tangent_basis of +z)
        const float n_w[3] = {0.0f, 0.0f, 1.0f};
        float t1[3], t2[3];
        tangent_basis(n_w, t1, t2);
#pragma unroll
        for (int i = 0; i < 3; ++i) F[i] = FMA(t2[i], c.lam[2], FMA(t1[i], c.lam[1], n_w[i] * c.lam[0])) * inv_h;
    }
    float T[3];
    cross3(c.arm, F, T);
#pragma unroll
    for (int i = 0; i < 3; ++i) { ft[i] += F[i]; ft[3 + i] += T[i]; }
}

// One solver substep of length h for the env held by this lane.  WRENCH: accumulate the fingertip contact
// wrench (only the asymmetric `states` vector consumes it).
template <bool WRENCH>
DEV void substep(const DevParams& P, Env& e, float h) {
    const TfModel& m = P.m;
    const float inv_h = 1.0f / h;
    // per-env cube and friction parameters: nominal values times the domain-randomisation factors (1.0 when off)
    const float cube_mass = m.cube_mass * e.dr[0];
    const float cube_inertia = m.cube_inertia * e.dr[0] * e.dr[1] * e.dr[1];
    const float inv_m = 1.0f / cube_mass, inv_I = 1.0f / cube_inertia;
    const float mu_fc = m.mu_finger_cube * e.dr[2], mu_tf = m.mu_tip_floor * e.dr[2];
    const float mu_cf = m.mu_cube_floor * e.dr[2], mu_cw = m.mu_cube_wall * e.dr[2];
    FK fk0, fk1, fk2;
    float vq[9];
    float v[3], w[3];
    // ---- free motion ----
    {
        float damp = 1.0f - h * m.link_angular_damping;
#define FREE_MOTION(F, fk)                                                                       \
        {                                                                                        \
            float M[6], bias[3], rhs[3], acc[3];                                                 \
            fk_setup(m, &e.q[3 * F], fk);                                                        \
            finger_dynamics(m, fk, &e.qd[3 * F], P.grav, M, bias);                               \
            for (int j = 0; j < 6; ++j) M[j] = M[j] * e.dr[4];    /* link-mass factor (1.0 when off) */ \
            for (int j = 0; j < 3; ++j) bias[j] = bias[j] * e.dr[4];                             \
            inv3sym(M, fk.Minv);                                                                 \
            for (int j = 0; j < 3; ++j) rhs[j] = e.tau[3 * F + j] - bias[j];                     \
            sym3_mul(fk.Minv, rhs, acc);                                                         \
            for (int j = 0; j < 3; ++j) vq[3 * F + j] = FMA(h, acc[j], e.qd[3 * F + j]) * damp;  \
        }
        FREE_MOTION(0, fk0)
        FREE_MOTION(1, fk1)
        FREE_MOTION(2, fk2)
#undef FREE_MOTION
        float dl = 1.0f - h * m.cube_linear_damping, da = 1.0f - h * m.cube_angular_damping;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            v[i] = FMA(h, P.grav[i], e.cv[i]) * dl;
            w[i] = e.cw[i] * da;
        }
    }
    PHASE_STAMP();
    // ---- contact generation ----
    float R[9];
    quat_to_rot(e.cq, R);
    const float hc = m.cube_half * e.dr[1];
    FingerContactRegs fc0, fc1, fc2, tf0, tf1, tf2;
    finger_contacts<0>(P, e, fk0, R, vq, v, w, hc, inv_h, inv_m, inv_I, fc0, tf0);
    finger_contacts<1>(P, e, fk1, R, vq, v, w, hc, inv_h, inv_m, inv_I, fc1, tf1);
    finger_contacts<2>(P, e, fk2, R, vq, v, w, hc, inv_h, inv_m, inv_I, fc2, tf2);
    PHASE_STAMP();
    CubeContactRegs cf[4], cwl[4];
    {   // cube vs floor: corners of the face that points down most
        int k = 0;
        float best = f_abs(R[6]);
        if (f_abs(R[7]) > best) { best = f_abs(R[7]); k = 1; }
        if (f_abs(R[8]) > best) { best = f_abs(R[8]); k = 2; }
        float rk = (k == 0) ? R[6] : ((k == 1) ? R[7] : R[8]);
        float sk = (rk > 0.0f) ? -1.0f : 1.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            CubeContactRegs& c = cf[i];
            cube_contact_zero(c);
            cube_corner(R, hc, k, sk, i, c.r);
            float gap = e.cp[2] + c.r[2];
            if (__builtin_expect(gap < m.contact_margin, 1)) {
                const float* r = c.r;
                c.Dinv[0] = f_rcp2(FMA(FMA(r[0], r[0], r[1] * r[1]), inv_I, inv_m));
                c.Dinv[1] = f_rcp2(FMA(FMA(r[2], r[2], r[1] * r[1]), inv_I, inv_m));
                c.Dinv[2] = f_rcp2(FMA(FMA(r[2], r[2], r[0] * r[0]), inv_I, inv_m));
                float vn0 = FMA(r[1], w[0], FMA(-r[0], w[1], v[2]));
                c.bias = contact_bias(m, gap, vn0, inv_h, 0.0f);
            }
        }
    }
    PHASE_STAMP();
    {   // cube vs boundary wall: corners of the face that points outward most
        float rc2 = FMA(e.cp[0], e.cp[0], e.cp[1] * e.cp[1]);
        float irc = f_rsqrt(f_max(rc2, 1e-24f));
        float rho_c = rc2 * irc;
        bool any = rho_c > 1e-6f;
        float dx = 0.0f, dy = 0.0f;
        if (any) { dx = e.cp[0] * irc; dy = e.cp[1] * irc; }
        float pr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pr[i] = FMA(R[i], dx, R[3 + i] * dy);
        int k = 0;
        float best = f_abs(pr[0]);
        if (f_abs(pr[1]) > best) { best = f_abs(pr[1]); k = 1; }
        if (f_abs(pr[2]) > best) { best = f_abs(pr[2]); k = 2; }
        float pk = (k == 0) ? pr[0] : ((k == 1) ? pr[1] : pr[2]);
        float sk = (pk < 0.0f) ? -1.0f : 1.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            CubeContactRegs& c = cwl[i];
            cube_contact_zero(c);
            cube_corner(R, hc, k, sk, i, c.r);
            float px = e.cp[0] + c.r[0], py = e.cp[1] + c.r[1];
            float rho2 = FMA(px, px, py * py);
            float inv = f_rsqrt(f_max(rho2, 1e-24f));
            float rho = rho2 * inv;
            float gap = m.wall_radius - rho;
            if (__builtin_expect(any && gap < m.contact_margin && rho > 1e-6f, 1)) {
                const float* r = c.r;
                c.n[0] = -px * inv; c.n[1] = -py * inv;
                float a[3], b[3];
                wall_arm_n(c, a);
                wall_arm_t(c, b);
                c.Dinv[0] = f_rcp2(FMA(dot3(a, a), inv_I, inv_m));
                c.Dinv[1] = f_rcp2(FMA(dot3(b, b), inv_I, inv_m));
                c.Dinv[2] = f_rcp2(FMA(FMA(r[0], r[0], r[1] * r[1]), inv_I, inv_m));
                float vn0 = FMA(a[2], w[2], FMA(a[1], w[1], FMA(a[0], w[0], FMA(c.n[1], v[1], c.n[0] * v[0]))));
                c.bias = contact_bias(m, gap, vn0, inv_h, 0.0f);
            }
        }
    }
    PHASE_STAMP();
    // ---- joint / velocity limit rows ----
    float vlo[9], vhi[9], lim_dinv[9], lim_lam[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int f = j / 3, jj = j % 3;
        const int dg = (jj == 0) ? 0 : ((jj == 1) ? 3 : 5);
        const FK& k = (f == 0) ? fk0 : ((f == 1) ? fk1 : fk2);
        vlo[j] = f_clamp((m.q_lo[jj] - e.q[j]) * inv_h, -m.qd_max, m.qd_max);
        vhi[j] = f_clamp((m.q_hi[jj] - e.q[j]) * inv_h, -m.qd_max, m.qd_max);
        lim_dinv[j] = f_rcp(k.Minv[dg]);
        lim_lam[j] = 0.0f;
    }
    PHASE_STAMP();
    // ---- projected Gauss-Seidel ----
    for (int it = 0; it < P.iters; ++it) {
        solve_finger_cube<0>(mu_fc, fc0, vq, v, w, inv_m, inv_I);
        solve_finger_cube<1>(mu_fc, fc1, vq, v, w, inv_m, inv_I);
        solve_finger_cube<2>(mu_fc, fc2, vq, v, w, inv_m, inv_I);
        solve_tip_floor<0>(mu_tf, tf0, vq);
        solve_tip_floor<1>(mu_tf, tf1, vq);
        solve_tip_floor<2>(mu_tf, tf2, vq);
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // cube - floor: rows +z (normal), +x, +y
            cube_row_z<0, true>(cf[i], mu_cf, inv_m, inv_I, v, w);
            cube_row_x<1>(cf[i], mu_cf, inv_m, inv_I, v, w);
            cube_row_y<2>(cf[i], mu_cf, inv_m, inv_I, v, w);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // cube - wall: rows n (normal), t, +z
            wall_row_n(cwl[i], inv_m, inv_I, v, w);
            wall_row_t(cwl[i], mu_cw, inv_m, inv_I, v, w);
            cube_row_z<2, false>(cwl[i], mu_cw, inv_m, inv_I, v, w);
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {       // joint limits + velocity limit
            const int f = j / 3, jj = j % 3;
            const FK& k = (f == 0) ? fk0 : ((f == 1) ? fk1 : fk2);
            const int dg = (jj == 0) ? 0 : ((jj == 1) ? 3 : 5);
            const int c0 = (jj == 0) ? 0 : ((jj == 1) ? 1 : 2);
            const int c1 = (jj == 0) ? 1 : ((jj == 1) ? 3 : 4);
            const int c2 = (jj == 0) ? 2 : ((jj == 1) ? 4 : 5);
            float v0 = FMA(-k.Minv[dg], lim_lam[j], vq[j]);
            float tgt = f_clamp(v0, vlo[j], vhi[j]);
            float lam_new = (tgt - v0) * lim_dinv[j];
            float dl = lam_new - lim_lam[j];
            lim_lam[j] = lam_new;
            vq[3 * f + 0] = FMA(k.Minv[c0], dl, vq[3 * f + 0]);
            vq[3 * f + 1] = FMA(k.Minv[c1], dl, vq[3 * f + 1]);
            vq[3 * f + 2] = FMA(k.Minv[c2], dl, vq[3 * f + 2]);
        }
    }
    PHASE_STAMP();
    // ---- fingertip wrench sensor ----
    if (WRENCH) {
        add_wrench<true>(fc0, inv_h, &e.ft[0]);
        add_wrench<false>(tf0, inv_h, &e.ft[0]);
        add_wrench<true>(fc1, inv_h, &e.ft[6]);
        add_wrench<false>(tf1, inv_h, &e.ft[6]);
        add_wrench<true>(fc2, inv_h, &e.ft[12]);
        add_wrench<false>(tf2, inv_h, &e.ft[12]);
    }
    // ---- integrate ----
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        e.qd[j] = vq[j];
        e.q[j] = f_clamp(FMA(h, vq[j], e.q[j]), m.q_lo[j % 3], m.q_hi[j % 3]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        e.cv[i] = v[i]; e.cw[i] = w[i];
        e.cp[i] = FMA(h, v[i], e.cp[i]);
    }
    quat_integrate(e.cq, e.cw, h);
    PHASE_STAMP();
}

// the moving goal (goal_movement.rotation) is a free body nothing interacts with: its orientation is
// integrated with the same substep sequence, outside the contact solve
DEV void goal_advance(const DevParams& P, Env& e, int nsub, float h) {
    if (P.goal_rotation_activate) {
        for (int s = 0; s < nsub; ++s) quat_integrate(e.gq, e.gw, h);
    }
}

// ------------------------------------------------------------------------------------------------------
// SoA <-> registers
// ------------------------------------------------------------------------------------------------------
// The state block float[88][N] is addressed through a raw buffer resource: wave-uniform row offset in an SGPR (soffset),
// 32-bit lane offset in one VGPR - buffer_load/store_dword ... offen, no per-lane 64-bit address arithmetic at all.
// (tf_create limits N so that 88*N*4 fits the 32-bit offsets.)
#define ST_RSRC() __builtin_amdgcn_make_buffer_rsrc((void*)P.state, 0, TF_STATE_ROWS * P.N * 4, 0x00020000)
#define LDST(row) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ST_RSRC(), (unsigned)i * 4u, (row) * P.N * 4, 0))
#define STST(row, val) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(val)), ST_RSRC(), (unsigned)i * 4u, (row) * P.N * 4, 0)

DEV void load_goal(const DevParams& P, int i, Env& e) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { e.gp[j] = LDST(TF_S_GOAL_P + j); e.gw[j] = LDST(TF_S_GOAL_W + j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) e.gq[j] = LDST(TF_S_GOAL_Q + j);
}
DEV void store_goal(const DevParams& P, int i, const Env& e, bool pred) {
    if (!pred) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) { STST(TF_S_GOAL_P + j, e.gp[j]); STST(TF_S_GOAL_W + j, e.gw[j]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) STST(TF_S_GOAL_Q + j, e.gq[j]);
}
DEV void load_dyn(const DevParams& P, int i, Env& e) {
#pragma unroll
    for (int j = 0; j < TF_NUM_DR; ++j) e.dr[j] = LDST(TF_S_DR + j);
#pragma unroll
    for (int j = 0; j < 9; ++j) { e.q[j] = LDST(TF_S_Q + j); e.qd[j] = LDST(TF_S_QD + j); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { e.cp[j] = LDST(TF_S_CUBE_P + j); e.cv[j] = LDST(TF_S_CUBE_V + j); e.cw[j] = LDST(TF_S_CUBE_W + j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) e.cq[j] = LDST(TF_S_CUBE_Q + j);
}
DEV void store_dyn(const DevParams& P, int i, const Env& e, bool valid) {
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < 9; ++j) { STST(TF_S_Q + j, e.q[j]); STST(TF_S_QD + j, e.qd[j]); STST(TF_S_TAU + j, e.tau[j]); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { STST(TF_S_CUBE_P + j, e.cp[j]); STST(TF_S_CUBE_V + j, e.cv[j]); STST(TF_S_CUBE_W + j, e.cw[j]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) STST(TF_S_CUBE_Q + j, e.cq[j]);
}
DEV void load_split_extras(const DevParams& P, int i, Env& e) {
#pragma unroll
    for (int j = 0; j < 9; ++j) e.tau[j] = LDST(TF_S_TAU + j);
#pragma unroll
    for (int j = 0; j < 18; ++j) e.ft[j] = LDST(TF_S_FT + j);
}
DEV void store_ft(const DevParams& P, int i, const Env& e, bool valid) {
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < 18; ++j) STST(TF_S_FT + j, e.ft[j]);
}
DEV void store_prev_obj(const DevParams& P, int i, const Env& e, bool valid) {
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) STST(TF_S_PREV_OBJ_P + j, e.cp[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) STST(TF_S_PREV_OBJ_Q + j, e.cq[j]);
}
DEV void load_prev_obj(const DevParams& P, int i, float prev_obj[7]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) prev_obj[j] = LDST(TF_S_PREV_OBJ_P + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) prev_obj[3 + j] = LDST(TF_S_PREV_OBJ_Q + j);
}

// ------------------------------------------------------------------------------------------------------
// task layer
// ------------------------------------------------------------------------------------------------------
DEV void sample_xy(float u_r, float u_t, float r_max, float& x, float& y) {   // reference sample.py:22-34
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

#define CUBE_RADIUS_3D 0.05629165f      // CuboidalObject(0.065).radius_3d, reference envs/trifinger/utils.py:122-131
#define CUBE_MAX_COM_DIST 0.13870835f
#define CUBE_MIN_HEIGHT 0.0325f

DEV void sample_goal(const DevParams& P, uint32_t gid, uint32_t count, Env& e) {   // trifinger_env.py:1194-1265
    int d = P.task_difficulty;
    float u[4];
    rng4(P, gid, count, RNG_GOAL_POS, u);
    float x = 0.0f, y = 0.0f, z;
    float quat[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    if (d == -1 || d == 1 || d == 3 || d == 4 || d == 5) sample_xy(u[0], u[1], CUBE_MAX_COM_DIST, x, y);
    if (d == -1 || d == 1) z = CUBE_MIN_HEIGHT;
    else if (d == 2 || d == 6) z = CUBE_MIN_HEIGHT + 0.05f;
    else if (d == 3) z = 0.0675f * u[2] + CUBE_MIN_HEIGHT;
    else z = 0.04370835f * u[2] + CUBE_RADIUS_3D;
    if (d == -1) sample_yaw_quat(u[3], quat);
    if (d == 4 || d == 5 || d == 6) {
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
        for (int i = 0; i < 3; ++i) e.gw[i] = mag * (n[i] / nrm);
    } else {
        e.gw[0] = 0.0f; e.gw[1] = 0.0f; e.gw[2] = 0.0f;
    }
    e.gp[0] = x; e.gp[1] = y; e.gp[2] = z;
#pragma unroll
    for (int i = 0; i < 4; ++i) e.gq[i] = quat[i];
}

// per-env values read once at the top of the step and carried in registers to the bookkeeping at its end
// (so that nothing at the end of the kernel waits for a global-memory round trip)
struct Carried {
    float tip_prev[9];       // history[0] fingertip positions of the previous frame
    bool successes;          // _successes after the reset logic
    bool goal_reset;         // _goal_reset_buf after the reset logic
    bool reset;              // _reset_buf after the reset logic
    int steps;               // _steps_count_buf after the reset logic
};

// flag and counter buffers of one env, fetched together with the state rows (one memory round trip for everything)
struct EnvFlags {
    uint8_t reset, goal_reset, successes;
    int steps;
    uint32_t count;
};
DEV void load_flags(const DevParams& P, int i, EnvFlags& fl) {
    fl.reset = P.reset_buf[(unsigned)i];
    fl.goal_reset = P.goal_reset_buf[(unsigned)i];
    fl.successes = P.successes[(unsigned)i];
    fl.steps = P.steps[(unsigned)i];
    fl.count = P.reset_count[(unsigned)i];
}

// masked _reset_impl then _goal_reset_impl (env_base.py:370-379; trifinger_env.py:373-440)
DEV bool apply_resets(const DevParams& P, int i, bool valid, Env& e, bool force_all, bool& goal_changed, Carried& cy,
                      const EnvFlags& fl) {
    const TfModel& m = P.m;
    uint32_t gid = (uint32_t)(P.env_id_offset + i);
    bool did_reset = false;
    bool rflag = force_all || (fl.reset != 0);
    bool gflag = !force_all && (fl.goal_reset != 0);
    uint32_t count = fl.count;
    if (rflag) {
        did_reset = true;
        if (P.dr_enable) {      // build-defined domain randomisation: scale = lo + (hi - lo) u
            float u[4];
            rng4(P, gid, count, RNG_DR, u);
            e.dr[0] = FMA(P.dr_cube_mass[1] - P.dr_cube_mass[0], u[0], P.dr_cube_mass[0]);
            e.dr[1] = FMA(P.dr_cube_size[1] - P.dr_cube_size[0], u[1], P.dr_cube_size[0]);
            e.dr[2] = FMA(P.dr_friction[1] - P.dr_friction[0], u[2], P.dr_friction[0]);
            e.dr[3] = FMA(P.dr_motor[1] - P.dr_motor[0], u[3], P.dr_motor[0]);
            rng4(P, gid, count, RNG_DR + 1u, u);
            e.dr[4] = FMA(P.dr_link_mass[1] - P.dr_link_mass[0], u[0], P.dr_link_mass[0]);
            e.dr[5] = FMA(P.dr_restitution[1] - P.dr_restitution[0], u[1], P.dr_restitution[0]);
        }
        if (P.robot_reset_type == TF_RESET_DEFAULT) {
#pragma unroll
            for (int j = 0; j < 9; ++j) { e.q[j] = m.q_default[j % 3]; e.qd[j] = 0.0f; }
        } else if (P.robot_reset_type == TF_RESET_RANDOM) {
            float n[20];
#pragma unroll
            for (int b = 0; b < 5; ++b) rng4(P, gid, count, RNG_ROBOT + (uint32_t)b, &n[4 * b]);
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                e.q[j] = m.q_default[j % 3] + P.dof_pos_stddev * (2.0f * n[j] - 1.0f);
                e.qd[j] = 0.0f + P.dof_vel_stddev * (2.0f * n[9 + j] - 1.0f);
            }
        }
        if (P.object_reset_type == TF_RESET_DEFAULT) {
            e.cp[0] = 0.0f; e.cp[1] = 0.0f; e.cp[2] = CUBE_MIN_HEIGHT * e.dr[1];
            e.cq[0] = 0.0f; e.cq[1] = 0.0f; e.cq[2] = 0.0f; e.cq[3] = 1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { e.cv[k] = 0.0f; e.cw[k] = 0.0f; }
        } else if (P.object_reset_type == TF_RESET_RANDOM) {
            float u[4];
            rng4(P, gid, count, RNG_OBJECT, u);
            sample_xy(u[0], u[1], CUBE_MAX_COM_DIST, e.cp[0], e.cp[1]);
            e.cp[2] = (0.065f / 2.0f) * e.dr[1];
            sample_yaw_quat(u[2], e.cq);
#pragma unroll
            for (int k = 0; k < 3; ++k) { e.cv[k] = 0.0f; e.cw[k] = 0.0f; }
        }
        sample_goal(P, gid, count, e);
        count = count + 1u;
    }
    if (gflag) {
        sample_goal(P, gid, count, e);
        count = count + 1u;
    }
    if (valid) {
        if (rflag) { P.reset_buf[(unsigned)i] = 0; P.steps[(unsigned)i] = 0; P.successes[(unsigned)i] = 0; }
        if (gflag) P.goal_reset_buf[(unsigned)i] = 0;
        if (rflag || gflag) P.reset_count[(unsigned)i] = count;
        if (rflag && P.dr_enable) {
#pragma unroll
            for (int j = 0; j < TF_NUM_DR; ++j) STST(TF_S_DR + j, e.dr[j]);
        }
    }
    goal_changed = rflag || gflag;
    cy.reset = false;                                   // cleared by the reset, or it was not set
    cy.goal_reset = force_all ? (fl.goal_reset != 0) : false;   // reset() leaves _goal_reset_buf alone
    cy.successes = rflag ? false : (fl.successes != 0);
    cy.steps = rflag ? 0 : fl.steps;
    return did_reset;
}

// trifinger_env.py:442-494
template <int A>
DEV void compute_torque(const DevParams& P, const float* act, const float q[9], const float qd[9], float motor_scale,
                        float tau[9]) {
    // the mode switches are wave-uniform: they are taken once around the joint loops (not once per joint), so that the
    // scalar table loads of a loop sit in one block and are fetched as one batch
    float at[A];
    if (P.normalize_action) {
#pragma unroll
        for (int j = 0; j < A; ++j) {
            float lo = P.tables[TAB_ACT_LO + j], hi = P.tables[TAB_ACT_HI + j];
            float off = (lo + hi) * 0.5f;
            at[j] = act[j] * (hi - lo) * 0.5f + off;
        }
    } else {
#pragma unroll
        for (int j = 0; j < A; ++j) at[j] = act[j];
    }
    float t[9];
    if (P.command_mode == TF_CMD_TORQUE) {
#pragma unroll
        for (int j = 0; j < 9; ++j) t[j] = at[j];
    } else if (P.command_mode == TF_CMD_POSITION) {
#pragma unroll
        for (int j = 0; j < 9; ++j) { t[j] = P.tables[TAB_KP + j] * (at[j] - q[j]); t[j] = t[j] - P.tables[TAB_KD + j] * qd[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 9; ++j) { t[j] = at[(A == 18) ? 9 + j : j] * (at[j] - q[j]); t[j] = t[j] - P.tables[TAB_KD + j] * qd[j]; }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] = f_max(f_min(t[j], 0.36f), -0.36f);
    if (P.apply_safety_damping) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            t[j] = t[j] - P.tables[TAB_KS + j] * qd[j];
            t[j] = f_max(f_min(t[j], 0.36f), -0.36f);
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) tau[j] = t[j] * motor_scale;     // domain randomisation of the motor strength (1.0 when off)
}

// build-defined action repeat: keep the previous step's applied torque (state rows TF_S_TAU, cleared by a reset) with
// probability dr_action_repeat.  The switch is wave-uniform: nothing is loaded or drawn when it is off.
template <int A>
DEV void torque_with_repeat(const DevParams& P, int i, uint32_t frame0, bool was_reset, const float* act, Env& e) {
    compute_torque<A>(P, act, e.q, e.qd, e.dr[3], e.tau);
    if (P.dr_action_repeat > 0.0f) {
        float u[4];
        rng4(P, (uint32_t)(P.env_id_offset + i), frame0, RNG_ACT_REPEAT, u);
        const bool keep = u[0] < P.dr_action_repeat;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float prev = was_reset ? 0.0f : LDST(TF_S_TAU + j);
            e.tau[j] = keep ? prev : e.tau[j];
        }
    }
}

DEV float norm3d(const float a[3], const float b[3]) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return f_sqrt(dx * dx + dy * dy + dz * dz);
}

template <int F> DEV void tip_state(const TfModel& m, const FK& k, const float q[3], const float qd[3], float out[13]) {
    float t[3], To[3];
    rot_link<3>(k, m.tip_origin, t);
    To[0] = k.p3[0] + t[0]; To[1] = k.p3[1] + t[1]; To[2] = k.p3[2] + t[2];
    base_to_world<F>(m, To, &out[0]);
    float sy, cy, sx, cx;
    tf_sincos(0.5f * q[0], sy, cy);
    tf_sincos(0.5f * (q[1] + q[2]), sx, cx);
    float qyx[4] = {cy * sx, sy * cx, -(sy * sx), cy * cx};
    float qz[4] = {0.0f, 0.0f, m.base_half_yaw_sin[F], m.base_half_yaw_cos[F]};
    quat_mul(qz, qyx, &out[3]);
    float L1[3], L2[3], L3[3], vb[3], wb[3];
    levers(k, To, L1, L2, L3);
#pragma unroll
    for (int i = 0; i < 3; ++i) vb[i] = L1[i] * qd[0] + L2[i] * qd[1] + L3[i] * qd[2];
    wb[0] = k.ax[0] * qd[1] + k.ax[0] * qd[2];
    wb[1] = qd[0];
    wb[2] = k.ax[2] * qd[1] + k.ax[2] * qd[2];
    dir_base_to_world<F>(m, vb, &out[7]);
    dir_base_to_world<F>(m, wb, &out[10]);
}

template <int F> DEV void wrench_local(const DevParams& P, const FK& kk, const Env& e, float inv_n, float out[6]) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float wv[3] = {e.ft[6 * F + 3 * half] * inv_n, e.ft[6 * F + 3 * half + 1] * inv_n, e.ft[6 * F + 3 * half + 2] * inv_n};
        float bv[3], lv[3];
        dir_world_to_base<F>(P.m, wv, bv);
        rot_link_T<3>(kk, bv, lv);
#pragma unroll
        for (int j = 0; j < 3; ++j) out[3 * half + j] = P.enable_ft ? lv[j] : 0.0f;
    }
}

// cooperative, coalesced store of a [64][W] tile staged in LDS as lds[lane * W + j]: ceil(16 W / 64) dwordx4 stores per
// lane, fully unrolled and branch-free - the tile is a raw buffer of total4*16 bytes, so the hardware range check drops
// the lanes past its end and all LDS reads (index clamped) can be in flight before the first store.  A ragged last
// wave (n_valid < 64) finishes its < 4 trailing floats with dword stores.
template <int W>
DEV void store_tile(gfloat* __restrict__ dst, const float* lds, int wave_first_env, int n_valid, int lane) {
    const unsigned total = (unsigned)(n_valid * W);              // floats in this wave's tile
    gfloat* base = dst + (size_t)wave_first_env * (size_t)W;     // 64*W*4-byte multiple: 16-B aligned
    const unsigned total4 = total >> 2;
    const __amdgpu_buffer_rsrc_t tile = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(total4 * 16u), 0x00020000);
    constexpr int ITER = (16 * W + WAVE - 1) / WAVE;
    const unsigned last4 = total4 - 1u;
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const unsigned idx = (unsigned)lane + (unsigned)(WAVE * k);
        const unsigned src = (idx < last4) ? idx : last4;
        u32x4 vv = *reinterpret_cast<const u32x4*>(&lds[src * 4u]);
        __builtin_amdgcn_raw_buffer_store_b128(vv, tile, idx * 16u, 0, 0);
    }
    const unsigned tail = (total4 << 2) + (unsigned)lane;
    if (tail < total) base[tail] = lds[tail];
}

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

// trifinger_env.py:500-559 + 959-1099 for the env of this lane.  prev_obj = history[1] pose (7).
template <int A>
DEV void post_step_env(const DevParams& P, const RewardCoef& rc, uint32_t frame, int i, bool valid, int wave_first, int n_valid, Env& e,
                       const float* act, const float prev_obj[7], bool with_reward, float* lds, int lane, LaneStats& st,
                       Carried& cy, StatsTicket& tk) {
    const TfModel& m = P.m;
    constexpr int OD = TF_OBS_DIM_BASE + A;
    constexpr int SD = OD + TF_STATES_EXTRA;
    float tips0[13], tips1[13], tips2[13];
    FK pk0, pk1, pk2;       // forward kinematics of the final pose: shared by the fingertip states and the wrench frames
    fk_setup(m, &e.q[0], pk0);
    fk_setup(m, &e.q[3], pk1);
    fk_setup(m, &e.q[6], pk2);
    tip_state<0>(m, pk0, &e.q[0], &e.qd[0], tips0);
    tip_state<1>(m, pk1, &e.q[3], &e.qd[3], tips1);
    tip_state<2>(m, pk2, &e.q[6], &e.qd[6], tips2);
    bool guarded = false;
    {   // NaN guard: a non-finite env is flagged for reset and parked at the default pose; its reward terms of this step
        // (they involve the histories that were non-finite) are zero: neither the learner nor the logged means see it
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc = acc + e.q[j] * 0.0f + e.qd[j] * 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc = acc + e.cp[j] * 0.0f + e.cv[j] * 0.0f + e.cw[j] * 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = acc + e.cq[j] * 0.0f;
        if (!(acc == 0.0f)) {
#pragma unroll
            for (int j = 0; j < 9; ++j) { e.q[j] = m.q_default[j % 3]; e.qd[j] = 0.0f; }
            e.cp[0] = 0.0f; e.cp[1] = 0.0f; e.cp[2] = CUBE_MIN_HEIGHT;
            e.cq[0] = 0.0f; e.cq[1] = 0.0f; e.cq[2] = 0.0f; e.cq[3] = 1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { e.cv[k] = 0.0f; e.cw[k] = 0.0f; }
#pragma unroll
            for (int j = 0; j < 18; ++j) e.ft[j] = 0.0f;
            fk_setup(m, &e.q[0], pk0);
            fk_setup(m, &e.q[3], pk1);
            fk_setup(m, &e.q[6], pk2);
            tip_state<0>(m, pk0, &e.q[0], &e.qd[0], tips0);
            tip_state<1>(m, pk1, &e.q[3], &e.qd[3], tips1);
            tip_state<2>(m, pk2, &e.q[6], &e.qd[6], tips2);
            if (valid) P.reset_buf[(unsigned)i] = 1;
            cy.reset = true;
            guarded = true;
            st.nonfinite += valid ? 1.0f : 0.0f;
        }
    }
    PHASE_STAMP();
    // ---- history: previous fingertip positions are whatever the last filled frame left ----
    const float* tip_prev = cy.tip_prev;
    if (valid) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { STST(TF_S_TIP_P + j, tips0[j]); STST(TF_S_TIP_P + 3 + j, tips1[j]); STST(TF_S_TIP_P + 6 + j, tips2[j]); }
    }
    // Rewards, termination and the episode statistics come BEFORE the observation tiles: the statistics atomic is in
    // flight while the tiles are emitted and stored (stats_end at the end of the kernel consumes its return).
    if (with_reward) {
        // ---- rewards (reference rewards.py; order of trifinger_env.py:513-550) ----
        float r[6];
        {
            float s = 0.0f;
            s = s + (norm3d(tips0, e.cp) - norm3d(&tip_prev[0], prev_obj));
            s = s + (norm3d(tips1, e.cp) - norm3d(&tip_prev[3], prev_obj));
            s = s + (norm3d(tips2, e.cp) - norm3d(&tip_prev[6], prev_obj));
            r[0] = rc.c_reach * s;
        }
        {
            float s = 0.0f;
    #pragma unroll
            for (int j = 0; j < 3; ++j) { float vel = (tips0[j] - tip_prev[j]) / rc.dt; s = s + vel * vel; }
    #pragma unroll
            for (int j = 0; j < 3; ++j) { float vel = (tips1[j] - tip_prev[3 + j]) / rc.dt; s = s + vel * vel; }
    #pragma unroll
            for (int j = 0; j < 3; ++j) { float vel = (tips2[j] - tip_prev[6 + j]) / rc.dt; s = s + vel * vel; }
            r[1] = rc.c_move_pen * s;
        }
        float dist = norm3d(e.cp, e.gp);
        r[2] = rc.c_dist * lgsk(dist, 50.0f);
        float ang = quat_diff_rad(e.cq, e.gq);
        r[3] = rc.w_rot * (rc.rot_num / (rc.rot_scale * f_abs(ang) + rc.rot_scale));
        float ang_prev = quat_diff_rad(&prev_obj[3], e.gq);
        r[4] = rc.w_rot_delta * (rc.rot_delta_sched * (f_abs(ang) - f_abs(ang_prev)));
        r[5] = rc.w_move * (dist - norm3d(prev_obj, e.gp));
        float total = 0.0f;
    #pragma unroll
        for (int t = 0; t < 6; ++t) {
            r[t] = guarded ? 0.0f : r[t];
            if (P.rew_active[t]) { total = total + r[t]; st.rew[t] += valid ? r[t] : 0.0f; }
        }
        // ---- termination (trifinger_env.py:1053-1099) ----
        bool pos_ok = dist <= P.pos_tol;
        bool ori_ok = ang <= P.ori_tol;
        st.pos_cnt += (valid && pos_ok) ? 1.0f : 0.0f;
        st.ori_cnt += (valid && ori_ok) ? 1.0f : 0.0f;
        bool done;
        if (P.task_difficulty < 4) done = pos_ok;
        else if (P.task_difficulty == 4) done = pos_ok && ori_ok;
        else done = ori_ok;
        bool succ = cy.successes;
        if (P.success_activate) {
            if (done) total = total + P.success_bonus;
            if (valid) P.goal_reset_buf[(unsigned)i] = (uint8_t)done;
            cy.goal_reset = done;
            succ = succ || done;
        } else {
            succ = cy.goal_reset && succ;
        }
        cy.successes = succ;
        if (valid) {
            P.successes[(unsigned)i] = (uint8_t)succ;
            P.reward[(unsigned)i] = total;
        }
        st.succ += (valid && succ) ? 1.0f : 0.0f;
    }
    stats_begin(P, st, lane, tk);
    PHASE_STAMP();
    // ---- observations: stage [lane][OD] in LDS, then one coalesced tile store ----
    const float* off = P.tables + TAB_OFF;
    const float* inv = P.tables + TAB_INV;
    const float co = P.clip_obs;        // fused wrapper clipping of every emitted value (FLT_MAX when off)
#define EMIT(W, col, val)                                                               \
    {                                                                                   \
        float x_ = (val);                                                               \
        lds[lane * (W) + (col)] = f_clamp(nrm ? FMA(x_, 2.0f * inv[col], -(2.0f * off[col]) * inv[col]) : x_, -co, co); \
    }
    // Every slot but the action one has limits that are constants of the MDP (reference trifinger_env.py:153-213):
    // with the loops unrolled, offset and 1/range fold into instruction literals - same fp32 values as the host-built
    // table ((lo+hi)*0.5f and 1.0f/(hi-lo)), no scalar loads.  The action slot depends on the configuration and
    // keeps reading the table.
#define EMITC(W, col, val, lo_, hi_)                                                    \
    {                                                                                   \
        float x_ = (val);                                                               \
        const float o_ = ((lo_) + (hi_)) * 0.5f, i_ = 1.0f / ((hi_) - (lo_));           \
        lds[lane * (W) + (col)] = f_clamp(nrm ? FMA(x_, 2.0f * i_, -(2.0f * o_) * i_) : x_, -co, co); \
    }
#define QLO(j) (((j) % 3 == 0) ? -0.33f : (((j) % 3 == 1) ? 0.0f : -2.7f))
#define QHI(j) (((j) % 3 == 0) ? 1.0f : (((j) % 3 == 1) ? 1.57f : 0.0f))
#define PLO(j) (((j) == 2) ? 0.0f : -0.3f)
#define TLO(j) (((j) < 2) ? -0.4f : (((j) == 2) ? 0.0f : (((j) < 7) ? -1.0f : -0.2f)))
#define THI(j) (((j) < 2) ? 0.4f : (((j) == 2) ? 0.5f : (((j) < 7) ? 1.0f : 0.2f)))
#define EMIT_COMMON(W)                                                                  \
    _Pragma("unroll") for (int j = 0; j < 9; ++j) EMITC(W, j, e.q[j], QLO(j), QHI(j))   \
    _Pragma("unroll") for (int j = 0; j < 9; ++j) EMITC(W, 9 + j, e.qd[j], -10.0f, 10.0f) \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) EMITC(W, 18 + j, e.cp[j], PLO(j), 0.3f) \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) EMITC(W, 21 + j, e.cq[j], -1.0f, 1.0f) \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) EMITC(W, 25 + j, e.gp[j], PLO(j), 0.3f) \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) EMITC(W, 28 + j, e.gq[j], -1.0f, 1.0f) \
    _Pragma("unroll") for (int j = 0; j < A; ++j) EMIT(W, 32 + j, act[j])
    // normalize_obs is wave-uniform: decided once around the whole emission (compile-time inside), so that the table
    // loads of the action slots sit in one block and nothing branches per element
    auto emit_tiles = [&](auto nrm_tag) {
    constexpr bool nrm = decltype(nrm_tag)::value;
    WAVE_LDS_ORDER();
    EMIT_COMMON(OD)
    if (P.dr_obs_noise > 0.0f) {        // wave-uniform; observation noise on q, qd and the object pose (build-defined DR)
        const uint32_t gid = (uint32_t)(P.env_id_offset + i);
        float nz[28];
#pragma unroll
        for (int b = 0; b < 7; ++b) rng4(P, gid, frame, RNG_OBS_NOISE + (uint32_t)b, &nz[4 * b]);
#pragma unroll
        for (int j = 0; j < 25; ++j) lds[lane * OD + j] = f_clamp(FMA(P.dr_obs_noise, 2.0f * nz[j] - 1.0f, lds[lane * OD + j]), -co, co);
    }
    WAVE_LDS_ORDER();
    store_tile<OD>(P.obs, lds, wave_first, n_valid, lane);
    WAVE_LDS_ORDER();
    PHASE_STAMP();
    if (P.asymmetric_obs) {
        EMIT_COMMON(SD)
#pragma unroll
        for (int j = 0; j < 3; ++j) EMITC(SD, OD + j, e.cv[j], -0.5f, 0.5f)
#pragma unroll
        for (int j = 0; j < 3; ++j) EMITC(SD, OD + 3 + j, e.cw[j], -0.5f, 0.5f)
#pragma unroll
        for (int j = 0; j < 13; ++j) EMITC(SD, OD + 6 + j, tips0[j], TLO(j), THI(j))
#pragma unroll
        for (int j = 0; j < 13; ++j) EMITC(SD, OD + 19 + j, tips1[j], TLO(j), THI(j))
#pragma unroll
        for (int j = 0; j < 13; ++j) EMITC(SD, OD + 32 + j, tips2[j], TLO(j), THI(j))
#pragma unroll
        for (int j = 0; j < 9; ++j) EMITC(SD, OD + 45 + j, (P.enable_ft ? e.tau[j] : 0.0f), -0.36f, 0.36f)
        float inv_n = 1.0f / (float)(P.substeps * P.control_decimation);
        float wl[6];
        wrench_local<0>(P, pk0, e, inv_n, wl);
#pragma unroll
        for (int j = 0; j < 6; ++j) EMITC(SD, OD + 54 + j, wl[j], -1.0f, 1.0f)
        wrench_local<1>(P, pk1, e, inv_n, wl);
#pragma unroll
        for (int j = 0; j < 6; ++j) EMITC(SD, OD + 60 + j, wl[j], -1.0f, 1.0f)
        wrench_local<2>(P, pk2, e, inv_n, wl);
#pragma unroll
        for (int j = 0; j < 6; ++j) EMITC(SD, OD + 66 + j, wl[j], -1.0f, 1.0f)
        WAVE_LDS_ORDER();
        store_tile<SD>(P.states, lds, wave_first, n_valid, lane);
        WAVE_LDS_ORDER();
    }
    };
    if (P.normalize_obs != 0) emit_tiles(std::true_type{}); else emit_tiles(std::false_type{});
#undef EMIT_COMMON
#undef EMIT
#undef EMITC
#undef QLO
#undef QHI
#undef PLO
#undef TLO
#undef THI
    PHASE_STAMP();
}

DEV void finish_env(const DevParams& P, int i, bool valid, const Carried& cy) {     // env_base.py:391-399
    if (!valid) return;
    int s = cy.steps + 1;
    P.steps[(unsigned)i] = s;
    bool rb = cy.reset;
    if (P.episode_length > 0 && s >= P.episode_length) { rb = true; P.reset_buf[(unsigned)i] = 1; }
    P.dones[(unsigned)i] = (uint8_t)(rb && cy.goal_reset);
}
DEV void load_carried(const DevParams& P, int i, Carried& cy) {        // split path: everything comes from memory
#pragma unroll
    for (int j = 0; j < 9; ++j) cy.tip_prev[j] = LDST(TF_S_TIP_P + j);
    cy.successes = P.successes[(unsigned)i] != 0;
    cy.goal_reset = P.goal_reset_buf[(unsigned)i] != 0;
    cy.reset = P.reset_buf[(unsigned)i] != 0;
    cy.steps = P.steps[(unsigned)i];
}

DEV void stats_zero(LaneStats& st) {
#pragma unroll
    for (int t = 0; t < 6; ++t) st.rew[t] = 0.0f;
    st.pos_cnt = 0.0f; st.ori_cnt = 0.0f; st.succ = 0.0f; st.resets = 0.0f; st.nonfinite = 0.0f;
}
// ------------------------------------------------------------------------------------------------------
// kernels.  One 64-lane wave per workgroup, one env per lane.  __launch_bounds__(64, 1): 1 wave/SIMD is
// all the chip ever holds at <= 65536 envs, so let the allocator use the whole VGPR file.
// Per-env data that is cold while the contact solve runs (goal pose, last action, previous object pose)
// is parked in its coalesced SoA rows in HBM/L2 and re-read afterwards, not carried in registers.
// ------------------------------------------------------------------------------------------------------
#define LANE_SETUP                                                     \
    const DevParams& P = *Pp;                                          \
    const int lane = threadIdx.x;                                      \
    const int wave_first = blockIdx.x * WAVE;                          \
    const int i_raw = wave_first + lane;                               \
    const bool valid = i_raw < P.N;                                    \
    const int i = valid ? i_raw : (P.N - 1);                           \
    const int n_valid = (P.N - wave_first < WAVE) ? (P.N - wave_first) : WAVE; \
    (void)n_valid; (void)i; (void)valid;

// fused control step (IS_RESET=false) or IsaacEnvBase.reset (IS_RESET=true)
template <int A, bool IS_RESET, bool ASYM>
__global__ void __launch_bounds__(WAVE, 1) k_step(const DevParams* __restrict__ Pp, const StepArgs sa,
                                                  const float* __restrict__ action) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    LANE_SETUP
    Env e;
    Carried cy;
    float n_resets = 0.0f;
    float act[A], prev_obj[7];
    PHASE_STAMP_RESET();
    PHASE_STAMP();
    // ---- phase A: action tile, masked resets, torque law ----
    {
        // every global load of the step is issued here, back to back (one exposed memory latency, not four): the action
        // tile [n_valid][A] (contiguous: coalesced dword loads, index clamped instead of branching), the state rows,
        // the flag / counter buffers
        float tile[A];
        EnvFlags fl;
        if (!IS_RESET) {
            const float* src = action + (size_t)wave_first * (size_t)A;
            const unsigned last = (unsigned)(n_valid * A) - 1u;
#pragma unroll
            for (int k = 0; k < A; ++k) {
                const unsigned idx = (unsigned)lane + (unsigned)(WAVE * k);
                tile[k] = src[idx < last ? idx : last];
            }
        }
        load_dyn(P, i, e);
        load_goal(P, i, e);
#pragma unroll
        for (int j = 0; j < 9; ++j) cy.tip_prev[j] = LDST(TF_S_TIP_P + j);
        load_flags(P, i, fl);
        PHASE_STAMP();
        if (!IS_RESET) {
            // the tile goes through LDS so that each lane can pick up its row
#pragma unroll
            for (int k = 0; k < A; ++k) lds[lane + WAVE * k] = tile[k];
            WAVE_LDS_ORDER();
            const int row = valid ? lane : (n_valid - 1);
#pragma unroll
            for (int j = 0; j < A; ++j) act[j] = f_clamp(lds[row * A + j], -P.clip_act, P.clip_act);
            WAVE_LDS_ORDER();
        } else {
#pragma unroll
            for (int j = 0; j < A; ++j) act[j] = 0.0f;
        }
        PHASE_STAMP();
        bool goal_changed;
        bool did_reset = apply_resets(P, i, valid, e, IS_RESET, goal_changed, cy, fl);
        PHASE_STAMP();
        store_goal(P, i, e, valid && goal_changed);
        if (did_reset) {
#pragma unroll
            for (int j = 0; j < A; ++j) act[j] = 0.0f;          // trifinger_env.py:387
            n_resets = valid ? 1.0f : 0.0f;
        }
        // _action_buf (what the observation reports as the last command): transposed back through LDS
#pragma unroll
        for (int j = 0; j < A; ++j) lds[lane * A + j] = act[j];
        WAVE_LDS_ORDER();
        store_tile<A>(P.action_buf, lds, wave_first, n_valid, lane);
        WAVE_LDS_ORDER();
        if (IS_RESET) compute_torque<A>(P, act, e.q, e.qd, e.dr[3], e.tau);
        else torque_with_repeat<A>(P, i, sa.frame0, did_reset, act, e);
        store_prev_obj(P, i, e, valid);                         // history[1] of the object (trifinger_env.py:975)
#pragma unroll
        for (int j = 0; j < 3; ++j) prev_obj[j] = e.cp[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) prev_obj[3 + j] = e.cq[j];
    }
    // Values that are cold while the contact solve runs (last action, goal, previous object pose and fingertips)
    // are parked in this wave's LDS tile buffer as [slot][lane] (idle between the two transposes) instead of
    // occupying 35+ registers through the substeps or being re-fetched from HBM at the end.
#define PARK(slot, val) lds[(slot) * WAVE + lane] = (val)
#define UNPARK(slot) lds[(slot) * WAVE + lane]
    {
        int sl = 0;
#pragma unroll
        for (int j = 0; j < A; ++j) PARK(sl++, act[j]);
#pragma unroll
        for (int j = 0; j < 7; ++j) PARK(sl++, prev_obj[j]);
#pragma unroll
        for (int j = 0; j < 9; ++j) PARK(sl++, cy.tip_prev[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) { PARK(sl++, e.gp[j]); PARK(sl++, e.gw[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) PARK(sl++, e.gq[j]);
    }
    // ---- phase B: physics ----
#pragma unroll
    for (int j = 0; j < 18; ++j) e.ft[j] = 0.0f;
    const int nsub = sa.nsim * P.substeps;
    PHASE_STAMP();
    for (int s = 0; s < nsub; ++s) substep<ASYM>(P, e, P.hsub);
    PHASE_STAMP();
    // ---- phase C: observations, rewards, termination, counters.  Its inputs come back from the LDS parking slots
    // (no global load sits between the last solver sweep and the output stores) ----
    WAVE_LDS_ORDER();
    {
        int sl = 0;
#pragma unroll
        for (int j = 0; j < A; ++j) act[j] = opaque(UNPARK(sl++));   // opaque: see DESIGN.md section 3, item 2
#pragma unroll
        for (int j = 0; j < 7; ++j) prev_obj[j] = UNPARK(sl++);
#pragma unroll
        for (int j = 0; j < 9; ++j) cy.tip_prev[j] = UNPARK(sl++);
#pragma unroll
        for (int j = 0; j < 3; ++j) { e.gp[j] = UNPARK(sl++); e.gw[j] = UNPARK(sl++); }
#pragma unroll
        for (int j = 0; j < 4; ++j) e.gq[j] = UNPARK(sl++);
    }
#undef PARK
#undef UNPARK
    WAVE_LDS_ORDER();
    {
        LaneStats st;
        stats_zero(st);
        st.resets = n_resets;
        goal_advance(P, e, nsub, P.hsub);
        StatsTicket tk;
        post_step_env<A>(P, sa.rc, sa.frame, i, valid, wave_first, n_valid, e, act, prev_obj, !IS_RESET, lds, lane, st, cy, tk);
        PHASE_STAMP();
        store_dyn(P, i, e, valid);
        if (P.goal_rotation_activate) store_goal(P, i, e, valid);
        if (!IS_RESET) finish_env(P, i, valid, cy);
        PHASE_STAMP();
        stats_end(P, lane, tk);
        PHASE_STAMP();
    }
}

// ---- split path: one hook per launch (parity tests) ----
template <int A>
__global__ void __launch_bounds__(WAVE, 1) k_apply_resets(const DevParams* __restrict__ Pp) {
    __shared__ __attribute__((aligned(16))) float lds[WAVE * 18];
    LANE_SETUP
    Env e;
    load_dyn(P, i, e);
    load_goal(P, i, e);
    bool goal_changed;
    Carried cy;
    EnvFlags fl;
    load_flags(P, i, fl);
    bool did = apply_resets(P, i, valid, e, false, goal_changed, cy, fl);
    float act[A];
    const int row = valid ? i : (P.N - 1);
#pragma unroll
    for (int j = 0; j < A; ++j) act[j] = did ? 0.0f : f_clamp(P.action_buf[(size_t)row * A + j], -P.clip_act, P.clip_act);
#pragma unroll
    for (int j = 0; j < A; ++j) lds[lane * A + j] = act[j];
    WAVE_LDS_ORDER();
    store_tile<A>(P.action_buf, lds, wave_first, n_valid, lane);
    load_split_extras(P, i, e);
    if (did) {
#pragma unroll
        for (int j = 0; j < 9; ++j) e.tau[j] = 0.0f;            // a reset clears the stored torque
    }
    store_dyn(P, i, e, valid);
    store_goal(P, i, e, valid);
}

template <int A>
__global__ void __launch_bounds__(WAVE, 1) k_pre_step(const DevParams* __restrict__ Pp, uint32_t frame0) {
    LANE_SETUP
    Env e;
    load_dyn(P, i, e);
    float act[A];
#pragma unroll
    for (int j = 0; j < A; ++j) act[j] = P.action_buf[(size_t)i * A + j];
    torque_with_repeat<A>(P, i, frame0, false, act, e);
#pragma unroll
    for (int j = 0; j < 18; ++j) e.ft[j] = 0.0f;
    store_dyn(P, i, e, valid);
    store_ft(P, i, e, valid);
    store_prev_obj(P, i, e, valid);
}

__global__ void __launch_bounds__(WAVE, 1) k_simulate(const DevParams* __restrict__ Pp) {
    LANE_SETUP
    Env e;
    load_dyn(P, i, e);
    load_split_extras(P, i, e);
    for (int s = 0; s < P.substeps; ++s) substep<true>(P, e, P.hsub);
    store_dyn(P, i, e, valid);
    store_ft(P, i, e, valid);
    if (P.goal_rotation_activate) {
        load_goal(P, i, e);
        goal_advance(P, e, P.substeps, P.hsub);
        store_goal(P, i, e, valid);
    }
}

template <int A>
__global__ void __launch_bounds__(WAVE, 1) k_post_step(const DevParams* __restrict__ Pp, const StepArgs sa) {
    __shared__ __attribute__((aligned(16))) float lds[WAVE * MAX_STATES];
    LANE_SETUP
    LaneStats st;
    stats_zero(st);
    Env e;
    load_dyn(P, i, e);
    load_goal(P, i, e);
    load_split_extras(P, i, e);
    float act[A];
#pragma unroll
    for (int j = 0; j < A; ++j) act[j] = P.action_buf[(size_t)i * A + j];
    float prev_obj[7];
    load_prev_obj(P, i, prev_obj);
    Carried cy;
    load_carried(P, i, cy);
    StatsTicket tk;
    post_step_env<A>(P, sa.rc, sa.frame, i, valid, wave_first, n_valid, e, act, prev_obj, true, lds, lane, st, cy, tk);
    store_dyn(P, i, e, valid);
    store_ft(P, i, e, valid);
    stats_end(P, lane, tk);
}

__global__ void __launch_bounds__(WAVE, 1) k_finish(const DevParams* __restrict__ Pp) {
    LANE_SETUP
    Carried cy;
    load_carried(P, i, cy);
    finish_env(P, i, valid, cy);
}

// ---- leaf kernels for the golden tests ----
__global__ void k_test_quat_diff(const float* a, const float* b, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float qa[4] = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
    float qb[4] = {b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]};
    out[i] = quat_diff_rad(qa, qb);
}
__global__ void k_test_quat_mul(const float* a, const float* b, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float qa[4] = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
    float qb[4] = {b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]};
    float o[4];
    quat_mul(qa, qb, o);
    for (int j = 0; j < 4; ++j) out[4 * i + j] = o[j];
}
__global__ void k_test_lgsk(const float* x, float scale, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = lgsk(x[i], scale);
}
__global__ void k_test_sample_xy(const float* ur, const float* ut, float r_max, float* x, float* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float xx, yy;
    sample_xy(ur[i], ut[i], r_max, xx, yy);
    x[i] = xx; y[i] = yy;
}
__global__ void k_test_yaw(const float* u, float* q, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[4];
    sample_yaw_quat(u[i], o);
    for (int j = 0; j < 4; ++j) q[4 * i + j] = o[j];
}
__global__ void k_test_normq(const float* nn, float* q, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float in[4] = {nn[4 * i], nn[4 * i + 1], nn[4 * i + 2], nn[4 * i + 3]}, o[4];
    normalize_quat(in, o);
    for (int j = 0; j < 4; ++j) q[4 * i + j] = o[j];
}
__global__ void k_test_philox(uint32_t k0, uint32_t k1, const uint32_t* env_id, const uint32_t* counter, uint32_t tag,
                              uint32_t* out4, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r[4];
    philox4x32_10(env_id[i], counter[i], tag, 0u, k0, k1, r);
    for (int j = 0; j < 4; ++j) out4[4 * i + j] = r[j];
}
__global__ void k_test_finger_dyn(const DevParams* __restrict__ Pp, const float* q, const float* qd, float* tip, float* mass, float* bias, int n) {
    const DevParams& P = *Pp;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FK k;
    float M[6], t[3], b[3];
    float qq[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]}, qv[3] = {qd[3 * i], qd[3 * i + 1], qd[3 * i + 2]};
    fk_setup(P.m, qq, k);
    finger_dynamics(P.m, k, qv, P.grav, M, b);
    rot_link<3>(k, P.m.tip_origin, t);
    for (int j = 0; j < 3; ++j) { tip[3 * i + j] = k.p3[j] + t[j]; bias[3 * i + j] = b[j]; }
    float* o = &mass[9 * i];
    o[0] = M[0]; o[1] = M[1]; o[2] = M[2]; o[3] = M[1]; o[4] = M[3]; o[5] = M[4]; o[6] = M[2]; o[7] = M[4]; o[8] = M[5];
}

// ======================================================================================================
// host side: handle + C ABI
// ======================================================================================================
struct TfHandle_ {
    TfConfig cfg;
    DevParams dp;            // host mirror of *d_params
    DevParams* d_params;     // device copy read by the kernels
    StepArgs sa;
    int bound;
    int64_t frame_count;
    int action_dim;
    // optional kernel timing (bench.py): event pairs around the fused step kernel
    hipEvent_t* ev;          // [2 * ev_cap]
    int ev_cap, ev_used;
    int ev_stride, ev_phase; // one event pair per window of ev_stride consecutive launches
};

static thread_local char g_err[512] = "";
static void free_events(TfHandle_* h);

static int hip_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return TF_ERR_DEVICE;
}
#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);    \
    } while (0)

extern "C" {

int tf_api_version(void) { return TF_API_VERSION; }
const char* tf_backend_name(void) { return "hip-gfx950"; }
const char* tf_last_error_string(void) { return g_err; }
int64_t tf_scratch_floats(int32_t num_envs) { return (int64_t)((num_envs + WAVE - 1) / WAVE) * SCR_STRIDE; }

int tf_action_dim(int32_t mode) {
    if (mode == TF_CMD_TORQUE || mode == TF_CMD_POSITION) return 9;
    if (mode == TF_CMD_POSITION_IMPEDANCE) return 18;
    return TF_ERR_COMMAND_MODE;
}

// Physical model from the URDF numbers (reference resources/assets/trifinger/robot_properties_fingers/urdf/pro/
// trifingerpro.urdf and objects/urdf/cube_multicolor_rrc.urdf).
void tf_default_model(TfModel* m) {
    memset(m, 0, sizeof(*m));
    m->base_height = 0.29f;
    const double yaw[3] = {0.0, -2.09439510239, -4.18879020479};
    for (int f = 0; f < 3; ++f) {
        m->base_yaw_cos[f] = (float)cos(yaw[f]);
        m->base_yaw_sin[f] = (float)sin(yaw[f]);
        m->base_half_yaw_cos[f] = (float)cos(0.5 * yaw[f]);
        m->base_half_yaw_sin[f] = (float)sin(0.5 * yaw[f]);
    }
    m->base_yaw_cos[0] = 1.0f; m->base_yaw_sin[0] = 0.0f;
    m->base_half_yaw_cos[0] = 1.0f; m->base_half_yaw_sin[0] = 0.0f;
    m->j2_origin[0] = 0.01685f; m->j2_origin[1] = 0.0505f; m->j2_origin[2] = 0.0f;
    m->j3_origin[0] = 0.04922f; m->j3_origin[1] = 0.0f;    m->j3_origin[2] = -0.16f;
    m->tip_origin[0] = 0.0185f; m->tip_origin[1] = 0.0f;   m->tip_origin[2] = -0.1626f;
    m->link_mass[0] = 0.26f;
    m->link_com[0][1] = 0.06f;
    m->link_inertia[0][0] = 0.000459333333333f; m->link_inertia[0][1] = 6.93333333333e-05f; m->link_inertia[0][2] = 0.000459333333333f;
    m->link_mass[1] = 0.25f;
    m->link_com[1][0] = 0.028f; m->link_com[1][2] = -0.08f;
    m->link_inertia[1][0] = 0.000441666666667f; m->link_inertia[1][1] = 0.000441666666667f; m->link_inertia[1][2] = 6.66666666667e-05f;
    {   // distal link: lower link merged with the rigidly attached tip link (parallel-axis, double precision)
        const double mass[2] = {0.021, 0.031};
        const double com[2][3] = {{0.0, 0.0, -0.06}, {0.0185, 0.0, -0.1626}};
        const double diag[2][3] = {{3.5e-05, 3.5e-05, 1.4e-06}, {5.16666666667e-07, 5.16666666667e-07, 5.16666666667e-07}};
        double mm = mass[0] + mass[1], c[3], I[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; ++i) c[i] = (mass[0] * com[0][i] + mass[1] * com[1][i]) / mm;
        I[0] = diag[0][0] + diag[1][0]; I[1] = diag[0][1] + diag[1][1]; I[2] = diag[0][2] + diag[1][2];
        for (int b = 0; b < 2; ++b) {
            double d[3] = {com[b][0] - c[0], com[b][1] - c[1], com[b][2] - c[2]};
            double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            I[0] += mass[b] * (d2 - d[0] * d[0]);
            I[1] += mass[b] * (d2 - d[1] * d[1]);
            I[2] += mass[b] * (d2 - d[2] * d[2]);
            I[3] += -mass[b] * d[0] * d[1];
            I[4] += -mass[b] * d[0] * d[2];
            I[5] += -mass[b] * d[1] * d[2];
        }
        m->link_mass[2] = (float)mm;
        for (int i = 0; i < 3; ++i) m->link_com[2][i] = (float)c[i];
        for (int i = 0; i < 6; ++i) m->link_inertia[2][i] = (float)I[i];
    }
    const float lo[3] = {-0.33f, 0.0f, -2.7f}, hi[3] = {1.0f, 1.57f, 0.0f}, df[3] = {0.0f, 0.9f, -1.7f};
    for (int i = 0; i < 3; ++i) { m->q_lo[i] = lo[i]; m->q_hi[i] = hi[i]; m->q_default[i] = df[i]; }
    m->qd_max = 10.0f;
    m->tau_max = 0.36f;
    m->link_angular_damping = 0.01f;
    m->cap_a[0] = 0.0135f;
    m->cap_b[0] = 0.0185f; m->cap_b[2] = -0.1592f;
    m->cap_radius = 0.0102f;
    m->cube_half = 0.0325f;
    m->cube_mass = (float)(291.3 * 0.065 * 0.065 * 0.065);
    m->cube_inertia = (float)(291.3 * 0.065 * 0.065 * 0.065 * 0.065 * 0.065 / 6.0);
    m->cube_linear_damping = 0.0f;
    m->cube_angular_damping = 0.05f;
    m->wall_radius = 0.192f;
    m->wall_height = 0.06f;
    m->mu_finger_cube = 1.0f;
    m->mu_cube_floor = 0.55f;
    m->mu_tip_floor = 0.55f;
    m->mu_cube_wall = 1.0f;
    m->restitution_finger = 0.4f;
    m->bounce_threshold = 0.5f;
    m->contact_margin = 0.04f;
    m->contact_offset = 0.002f;
    m->erp = 0.2f;
    m->max_depenetration_velocity = 1000.0f;
}

// scale tables: reference trifinger_env.py:153-213 (limits) and :655-710 (concatenation order)
static void build_tables(const TfConfig* c, int A, float* tab, int* obs_dim, int* states_dim) {
    const float q_lo[3] = {-0.33f, 0.0f, -2.7f}, q_hi[3] = {1.0f, 1.57f, 0.0f};
    const float kd[3] = {0.1f, 0.3f, 0.001f}, ks[3] = {0.08f, 0.08f, 0.04f};
    float lo[MAX_STATES], hi[MAX_STATES];
    float* act_lo = tab + TAB_ACT_LO;
    float* act_hi = tab + TAB_ACT_HI;
    for (int j = 0; j < 9; ++j) { tab[TAB_KP + j] = 10.0f; tab[TAB_KD + j] = kd[j % 3]; tab[TAB_KS + j] = ks[j % 3]; }
    for (int j = 0; j < 18; ++j) { act_lo[j] = 0.0f; act_hi[j] = 0.0f; }
    for (int j = 0; j < A; ++j) {
        if (c->command_mode == TF_CMD_TORQUE) { act_lo[j] = -0.36f; act_hi[j] = 0.36f; }
        else if (j < 9) { act_lo[j] = q_lo[j % 3]; act_hi[j] = q_hi[j % 3]; }
        else { act_lo[j] = 1.0f; act_hi[j] = 50.0f; }
    }
    int k = 0;
    for (int j = 0; j < 9; ++j) { lo[k] = q_lo[j % 3]; hi[k] = q_hi[j % 3]; ++k; }
    for (int j = 0; j < 9; ++j) { lo[k] = -10.0f; hi[k] = 10.0f; ++k; }
    for (int rep = 0; rep < 2; ++rep) {
        lo[k] = -0.3f; hi[k] = 0.3f; ++k; lo[k] = -0.3f; hi[k] = 0.3f; ++k; lo[k] = 0.0f; hi[k] = 0.3f; ++k;
        for (int j = 0; j < 4; ++j) { lo[k] = -1.0f; hi[k] = 1.0f; ++k; }
    }
    for (int j = 0; j < A; ++j) {
        if (c->normalize_action) { lo[k] = -1.0f; hi[k] = 1.0f; }
        else { lo[k] = act_lo[j]; hi[k] = act_hi[j]; }
        ++k;
    }
    *obs_dim = k;
    for (int j = 0; j < 6; ++j) { lo[k] = -0.5f; hi[k] = 0.5f; ++k; }
    for (int f = 0; f < 3; ++f) {
        lo[k] = -0.4f; hi[k] = 0.4f; ++k; lo[k] = -0.4f; hi[k] = 0.4f; ++k; lo[k] = 0.0f; hi[k] = 0.5f; ++k;
        for (int j = 0; j < 4; ++j) { lo[k] = -1.0f; hi[k] = 1.0f; ++k; }
        for (int j = 0; j < 6; ++j) { lo[k] = -0.2f; hi[k] = 0.2f; ++k; }
    }
    for (int j = 0; j < 9; ++j) { lo[k] = -0.36f; hi[k] = 0.36f; ++k; }
    for (int j = 0; j < 18; ++j) { lo[k] = -1.0f; hi[k] = 1.0f; ++k; }
    *states_dim = k;
    for (int j = 0; j < MAX_STATES; ++j) { tab[TAB_OFF + j] = 0.0f; tab[TAB_INV + j] = 1.0f; }
    for (int j = 0; j < k; ++j) {
        tab[TAB_OFF + j] = (lo[j] + hi[j]) * 0.5f;
        tab[TAB_INV + j] = 1.0f / (hi[j] - lo[j]);
    }
}

int tf_create(const TfConfig* cfg, tf_handle* out) {
    if (!cfg || !out) return TF_ERR_INVALID_ARG;
    if (cfg->api_version != TF_API_VERSION || cfg->num_envs <= 0) return TF_ERR_INVALID_ARG;
    if (cfg->num_envs > TF_MAX_ENVS) return TF_ERR_INVALID_ARG;     // 32-bit buffer offsets into state[88][N]
    if (tf_action_dim(cfg->command_mode) < 0) return TF_ERR_COMMAND_MODE;
    if (cfg->robot_reset_type < 0 || cfg->robot_reset_type > 2) return TF_ERR_ROBOT_RESET;
    if (cfg->object_reset_type < 0 || cfg->object_reset_type > 2) return TF_ERR_OBJECT_RESET;
    int d = cfg->task_difficulty;
    if (!(d == -1 || (d >= 1 && d <= 6))) return TF_ERR_DIFFICULTY;
    if (cfg->finger_reach_norm_p != 2) return TF_ERR_UNSUPPORTED;
    if (cfg->substeps <= 0 || cfg->solver_iterations <= 0 || cfg->control_decimation <= 0 || !(cfg->dt > 0.0f))
        return TF_ERR_INVALID_ARG;
    TfHandle_* h = new TfHandle_();
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->ev_stride = 1;
    if (h->cfg.global_num_envs <= 0) h->cfg.global_num_envs = cfg->num_envs;
    h->action_dim = tf_action_dim(cfg->command_mode);
    int od = 0, sd = 0;
    DevParams& P = h->dp;
    build_tables(&h->cfg, h->action_dim, P.tables, &od, &sd);
    hipError_t e;
    P.N = cfg->num_envs; P.A = h->action_dim; P.OD = od; P.SD = sd;
    P.env_id_offset = cfg->env_id_offset;
    P.seed_lo = (uint32_t)cfg->seed; P.seed_hi = (uint32_t)(cfg->seed >> 32);
    P.command_mode = cfg->command_mode; P.normalize_action = cfg->normalize_action; P.normalize_obs = cfg->normalize_obs;
    P.apply_safety_damping = cfg->apply_safety_damping; P.asymmetric_obs = cfg->asymmetric_obs; P.enable_ft = cfg->enable_ft_sensors;
    P.task_difficulty = cfg->task_difficulty; P.episode_length = cfg->episode_length;
    P.robot_reset_type = cfg->robot_reset_type; P.object_reset_type = cfg->object_reset_type;
    P.goal_rotation_activate = cfg->goal_rotation_activate;
    P.dr_enable = cfg->dr_enable;
    for (int i = 0; i < 2; ++i) {
        P.dr_cube_mass[i] = cfg->dr_cube_mass[i]; P.dr_cube_size[i] = cfg->dr_cube_size[i]; P.dr_friction[i] = cfg->dr_friction[i];
        P.dr_motor[i] = cfg->dr_motor[i]; P.dr_link_mass[i] = cfg->dr_link_mass[i]; P.dr_restitution[i] = cfg->dr_restitution[i];
    }
    P.dr_obs_noise = (cfg->dr_enable && cfg->dr_obs_noise > 0.0f) ? cfg->dr_obs_noise : 0.0f;
    P.dr_action_repeat = (cfg->dr_enable && cfg->dr_action_repeat > 0.0f) ? cfg->dr_action_repeat : 0.0f;
    P.clip_obs = 3.402823466e38f; P.clip_act = 3.402823466e38f;
    P.dof_pos_stddev = cfg->dof_pos_stddev; P.dof_vel_stddev = cfg->dof_vel_stddev; P.goal_rate = cfg->goal_rotation_rate_magnitude;
    for (int t = 0; t < 6; ++t) P.rew_active[t] = cfg->reward[t].activate;
    P.success_activate = cfg->success_activate; P.success_bonus = cfg->success_bonus;
    P.pos_tol = cfg->position_tolerance; P.ori_tol = cfg->orientation_tolerance;
    P.substeps = cfg->substeps; P.iters = cfg->solver_iterations; P.control_decimation = cfg->control_decimation;
    P.dt = cfg->dt; P.hsub = cfg->dt / (float)cfg->substeps;
    for (int i = 0; i < 3; ++i) P.grav[i] = cfg->gravity[i];
    P.m = cfg->model;
    void* tk = nullptr;
    e = hipMalloc(&tk, STAT_WORDS * sizeof(unsigned long long));
    if (e != hipSuccess) { delete h; return hip_fail(e, "hipMalloc(tickets)"); }
    P.tickets = (gu32*)tk;
    e = hipMemset(tk, 0, STAT_WORDS * sizeof(unsigned long long));
    if (e != hipSuccess) { (void)hipFree((void*)P.tickets); delete h; return hip_fail(e, "hipMemset(tickets)"); }
    e = hipMalloc((void**)&h->d_params, sizeof(DevParams));
    if (e != hipSuccess) { (void)hipFree((void*)P.tickets); delete h; return hip_fail(e, "hipMalloc(params)"); }
    e = hipMemcpy(h->d_params, &h->dp, sizeof(DevParams), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->d_params); delete h; return hip_fail(e, "hipMemcpy(params)"); }
    *out = h;
    return TF_OK;
}

int tf_destroy(tf_handle h) {
    if (!h) return TF_OK;
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->dp.tickets) (void)hipFree((void*)h->dp.tickets);
    free_events(h);
    delete h;
    return TF_OK;
}

int tf_bind(tf_handle h, const TfBuffers* b) {
    if (!h || !b) return TF_ERR_INVALID_ARG;
    if (!b->state || !b->action_buf || !b->obs || !b->reward || !b->reset_buf || !b->goal_reset_buf ||
        !b->successes || !b->dones || !b->steps || !b->reset_count || !b->info || !b->scratch)
        return TF_ERR_INVALID_ARG;
    if (h->cfg.asymmetric_obs && !b->states) return TF_ERR_INVALID_ARG;
    DevParams& P = h->dp;
    P.state = (gfloat*)b->state; P.action_buf = (gfloat*)b->action_buf; P.obs = (gfloat*)b->obs;
    P.states = (gfloat*)b->states; P.reward = (gfloat*)b->reward;
    P.reset_buf = (gu8*)b->reset_buf; P.goal_reset_buf = (gu8*)b->goal_reset_buf; P.successes = (gu8*)b->successes;
    P.dones = (gu8*)b->dones; P.steps = (gi32*)b->steps; P.reset_count = (gu32*)b->reset_count;
    P.info = (gfloat*)b->info; P.scratch = (gfloat*)b->scratch;
    HIP_TRY(hipMemcpy(h->d_params, &h->dp, sizeof(DevParams), hipMemcpyHostToDevice));
    h->bound = 1;
    return TF_OK;
}

int tf_set_clipping(tf_handle h, float clip_obs, float clip_actions) {
    if (!h) return TF_ERR_INVALID_ARG;
    h->dp.clip_obs = (clip_obs > 0.0f) ? clip_obs : 3.402823466e38f;
    h->dp.clip_act = (clip_actions > 0.0f) ? clip_actions : 3.402823466e38f;
    HIP_TRY(hipMemcpy(h->d_params, &h->dp, sizeof(DevParams), hipMemcpyHostToDevice));
    return TF_OK;
}
int tf_set_gravity(tf_handle h, const float g[3]) {
    if (!h || !g) return TF_ERR_INVALID_ARG;
    for (int i = 0; i < 3; ++i) { h->cfg.gravity[i] = g[i]; h->dp.grav[i] = g[i]; }
    // cold path: a blocking copy is fine (and orders after any step already queued on the null stream)
    HIP_TRY(hipMemcpy(h->d_params, &h->dp, sizeof(DevParams), hipMemcpyHostToDevice));
    return TF_OK;
}
int64_t tf_frame_count(tf_handle h) { return h ? h->frame_count : -1; }
int tf_set_frame_count(tf_handle h, int64_t f) { if (!h) return TF_ERR_INVALID_ARG; h->frame_count = f; return TF_OK; }

}  // extern "C"

static double sched_window(const TfRewardTerm* t, double step) {
    if (t->sched_start != t->sched_end) return (t->sched_start <= step && step <= t->sched_end) ? 1.0 : 0.0;
    return 1.0;
}
// scalar prefactors exactly as python evaluates them in double before they meet an fp32 tensor
// (reference rewards.py:50-63,117-139,165-184,203-235,245-263; env_base.py:287-289)
static void reward_coefs(TfHandle_* h) {
    const TfConfig* c = &h->cfg;
    RewardCoef* rc = &h->sa.rc;
    double step = (double)h->frame_count * (double)c->global_num_envs;
    double dt = (double)c->dt;
    const TfRewardTerm* T = c->reward;
    rc->c_reach = (float)((double)T[TF_REW_FINGER_REACH_OBJECT_RATE].weight * sched_window(&T[TF_REW_FINGER_REACH_OBJECT_RATE], step));
    rc->c_move_pen = T[TF_REW_FINGER_MOVE_PENALTY].weight;
    rc->dt = c->dt;
    rc->c_dist = (float)((double)T[TF_REW_OBJECT_DIST].weight * dt * sched_window(&T[TF_REW_OBJECT_DIST], step));
    rc->rot_num = (float)(sched_window(&T[TF_REW_OBJECT_ROT], step) * dt);
    rc->rot_scale = c->object_rot_scale;
    rc->w_rot = T[TF_REW_OBJECT_ROT].weight;
    const TfRewardTerm* t = &T[TF_REW_OBJECT_ROT_DELTA];
    double s = 1.0;
    if (t->sched_start != t->sched_end) {
        s = (step - t->sched_start) / (t->sched_end - t->sched_start);
        s = (s < 0.0) ? 0.0 : ((s > 1.0) ? 1.0 : s);
    }
    rc->rot_delta_sched = (float)s;
    rc->w_rot_delta = t->weight;
    rc->w_move = T[TF_REW_OBJECT_MOVE].weight;
}

static inline int n_waves(const TfHandle_* h) { return (h->cfg.num_envs + WAVE - 1) / WAVE; }

#define CHECK_HANDLE(h)                         \
    if (!(h)) return TF_ERR_INVALID_ARG;        \
    if (!(h)->bound) return TF_ERR_NOT_BOUND;

#define LAUNCH_CHECK(what)                                         \
    do {                                                           \
        hipError_t e_ = hipGetLastError();                         \
        if (e_ != hipSuccess) return hip_fail(e_, what);           \
    } while (0)

static int launch_step(TfHandle_* h, const float* action, bool is_reset, hipStream_t s) {
    const int nsim = is_reset ? 1 : h->cfg.control_decimation;
    h->frame_count += nsim;
    h->sa.nsim = nsim;
    h->sa.frame = (uint32_t)h->frame_count;
    h->sa.frame0 = (uint32_t)(h->frame_count - nsim);
    reward_coefs(h);
    dim3 grid(n_waves(h)), block(WAVE);
    // a full reset also re-arms the statistics accumulators (they are left at zero by every completed launch; this only
    // matters after a launch that did not complete)
    if (is_reset) HIP_TRY(hipMemsetAsync((void*)h->dp.tickets, 0, STAT_WORDS * sizeof(unsigned long long), s));
    const bool timing = !is_reset && h->ev && h->ev_used < h->ev_cap;
    if (timing && h->ev_phase == 0) HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used], s));      // window opens
    const bool asym = h->cfg.asymmetric_obs != 0;
#define LAUNCH_STEP(AA, RR, SS) hipLaunchKernelGGL((k_step<AA, RR, SS>), grid, block, 0, s, h->d_params, h->sa, action)
    if (h->action_dim == 9) {
        if (is_reset) { if (asym) LAUNCH_STEP(9, true, true); else LAUNCH_STEP(9, true, false); }
        else { if (asym) LAUNCH_STEP(9, false, true); else LAUNCH_STEP(9, false, false); }
    } else {
        if (is_reset) { if (asym) LAUNCH_STEP(18, true, true); else LAUNCH_STEP(18, true, false); }
        else { if (asym) LAUNCH_STEP(18, false, true); else LAUNCH_STEP(18, false, false); }
    }
#undef LAUNCH_STEP
    LAUNCH_CHECK("k_step");
    if (timing) {
        h->ev_phase += 1;
        if (h->ev_phase == h->ev_stride) {                                                   // window closes
            HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used + 1], s));
            h->ev_used += 1; h->ev_phase = 0;
        }
    }
    return TF_OK;
}

extern "C" {

int tf_step(tf_handle h, const float* action, void* stream) {
    CHECK_HANDLE(h)
    if (!action) return TF_ERR_INVALID_ARG;
    return launch_step(h, action, false, (hipStream_t)stream);
}
int tf_reset(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    return launch_step(h, nullptr, true, (hipStream_t)stream);
}

static void free_events(TfHandle_* h) {
    if (h->ev) {
        for (int i = 0; i < 2 * h->ev_cap; ++i) (void)hipEventDestroy(h->ev[i]);
        delete[] h->ev;
    }
    h->ev = nullptr; h->ev_cap = 0; h->ev_used = 0; h->ev_phase = 0;
}
int tf_enable_kernel_timing(tf_handle h, int32_t max_launches) {
    if (!h) return TF_ERR_INVALID_ARG;
    free_events(h);
    if (max_launches <= 0) return TF_OK;
    h->ev = new hipEvent_t[2 * (size_t)max_launches];
    for (int i = 0; i < 2 * max_launches; ++i) HIP_TRY(hipEventCreate(&h->ev[i]));
    h->ev_cap = max_launches;
    return TF_OK;
}
int tf_set_kernel_timing_window(tf_handle h, int32_t window) {
    if (!h || window <= 0) return TF_ERR_INVALID_ARG;
    h->ev_stride = window; h->ev_phase = 0;
    return TF_OK;
}
int tf_kernel_time_ms(tf_handle h, double* total_ms, int64_t* launches) {
    if (!h || !total_ms || !launches) return TF_ERR_INVALID_ARG;
    double sum = 0.0;
    for (int i = 0; i < h->ev_used; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(h->ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
        sum += (double)ms;
    }
    *total_ms = sum;
    *launches = (int64_t)h->ev_used * h->ev_stride;
    return TF_OK;
}

int tf_apply_resets(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    dim3 grid(n_waves(h)), block(WAVE);
    if (h->action_dim == 9) hipLaunchKernelGGL(k_apply_resets<9>, grid, block, 0, (hipStream_t)stream, h->d_params);
    else hipLaunchKernelGGL(k_apply_resets<18>, grid, block, 0, (hipStream_t)stream, h->d_params);
    LAUNCH_CHECK("k_apply_resets");
    return TF_OK;
}
int tf_pre_step(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    dim3 grid(n_waves(h)), block(WAVE);
    const uint32_t frame0 = (uint32_t)h->frame_count;
    if (h->action_dim == 9) hipLaunchKernelGGL(k_pre_step<9>, grid, block, 0, (hipStream_t)stream, h->d_params, frame0);
    else hipLaunchKernelGGL(k_pre_step<18>, grid, block, 0, (hipStream_t)stream, h->d_params, frame0);
    LAUNCH_CHECK("k_pre_step");
    return TF_OK;
}
int tf_simulate(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    h->frame_count += 1;
    hipLaunchKernelGGL(k_simulate, dim3(n_waves(h)), dim3(WAVE), 0, (hipStream_t)stream, h->d_params);
    LAUNCH_CHECK("k_simulate");
    return TF_OK;
}
int tf_post_step(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    reward_coefs(h);
    h->sa.frame = (uint32_t)h->frame_count;
    dim3 grid(n_waves(h)), block(WAVE);
    if (h->action_dim == 9) hipLaunchKernelGGL(k_post_step<9>, grid, block, 0, (hipStream_t)stream, h->d_params, h->sa);
    else hipLaunchKernelGGL(k_post_step<18>, grid, block, 0, (hipStream_t)stream, h->d_params, h->sa);
    LAUNCH_CHECK("k_post_step");
    return TF_OK;
}
int tf_finish_step(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    hipLaunchKernelGGL(k_finish, dim3(n_waves(h)), dim3(WAVE), 0, (hipStream_t)stream, h->d_params);
    LAUNCH_CHECK("k_finish");
    return TF_OK;
}

#define LEAF_GRID(n) dim3(((n) + 255) / 256), dim3(256), 0, (hipStream_t)stream
int tf_test_quat_diff_rad(const float* a, const float* b, float* out, int32_t n, void* stream) {
    hipLaunchKernelGGL(k_test_quat_diff, LEAF_GRID(n), a, b, out, n);
    LAUNCH_CHECK("k_test_quat_diff");
    return TF_OK;
}
int tf_test_quat_mul(const float* a, const float* b, float* out, int32_t n, void* stream) {
    hipLaunchKernelGGL(k_test_quat_mul, LEAF_GRID(n), a, b, out, n);
    LAUNCH_CHECK("k_test_quat_mul");
    return TF_OK;
}
int tf_test_lgsk(const float* x, float scale, float* out, int32_t n, void* stream) {
    hipLaunchKernelGGL(k_test_lgsk, LEAF_GRID(n), x, scale, out, n);
    LAUNCH_CHECK("k_test_lgsk");
    return TF_OK;
}
int tf_test_sample_xy(const float* ur, const float* ut, float r_max, float* x, float* y, int32_t n, void* stream) {
    hipLaunchKernelGGL(k_test_sample_xy, LEAF_GRID(n), ur, ut, r_max, x, y, n);
    LAUNCH_CHECK("k_test_sample_xy");
    return TF_OK;
}
int tf_test_sample_yaw_quat(const float* u, float* quat, int32_t n, void* stream) {
    hipLaunchKernelGGL(k_test_yaw, LEAF_GRID(n), u, quat, n);
    LAUNCH_CHECK("k_test_yaw");
    return TF_OK;
}
int tf_test_normalize_quat(const float* nn, float* quat, int32_t n, void* stream) {
    hipLaunchKernelGGL(k_test_normq, LEAF_GRID(n), nn, quat, n);
    LAUNCH_CHECK("k_test_normq");
    return TF_OK;
}
int tf_test_philox(uint64_t seed, const uint32_t* env_id, const uint32_t* counter, uint32_t tag, uint32_t* out4,
                   int32_t n, void* stream) {
    hipLaunchKernelGGL(k_test_philox, LEAF_GRID(n), (uint32_t)seed, (uint32_t)(seed >> 32), env_id, counter, tag, out4, n);
    LAUNCH_CHECK("k_test_philox");
    return TF_OK;
}
int tf_test_finger_dynamics(tf_handle h, const float* q, const float* qd, float* tip, float* mass, float* bias,
                            int32_t n, void* stream) {
    if (!h) return TF_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_test_finger_dyn, LEAF_GRID(n), h->d_params, q, qd, tip, mass, bias, n);
    LAUNCH_CHECK("k_test_finger_dyn");
    return TF_OK;
}

}  // extern "C"
