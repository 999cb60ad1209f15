/*
 * tf_oracle.c - TEST INFRASTRUCTURE.  Scalar C restatement of the TriFinger env-step hot path.
 *
 * This file is the ORACLE of the repository ("Oracle").  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from
 * it.  The product (leibnizgym_amd/csrc/trifinger_hip.hip) never includes, links or calls it.
 *
 * What it restates, and how each part is pinned:
 *   - task layer (reference arithmetic visible): leibnizgym/envs/env_base.py:345-401,
 *     leibnizgym/envs/trifinger/trifinger_env.py:373-559,959-1265, rewards.py, sample.py,
 *     leibnizgym/utils/torch_utils.py.  PINNED against tests/golden/{name}.npz, which were produced by
 *     importing the reference's own functions (tests/golden/make_golden.py).
 *   - physics (reference arithmetic NOT visible: it lives in the closed-source IsaacGym/PhysX binary,
 *     "NVIDIA IsaacGym Preview Release 2", README.md:17, not under /root/reference, no version pin):
 *     PARITY WITH ISAACGYM UNPINNED.  The algorithm below is this build's own spec (DESIGN.md section "Physics spec"),
 *     written from the URDF numbers and the solver settings the reference asks for.  The spec itself is pinned
 *     independently of this file: free motion against an fp64 Lagrangian model of the URDF chain
 *     (tests/test_physics_analytic.py), the contact solve against an fp64 restatement that shares no code with this
 *     file and iterates to a fixed point (tests/physics_ref.py, tests/test_contact_lcp_reference.py: 200 random contact
 *     configurations; tests/test_box_object.py: 60 with a general box and the full inertia tensor), and known-answer
 *     scenarios (tests/test_contact_scenarios.py: pinch-and-lift, impact, restitution, link / finger / wall contacts).
 *
 * Arithmetic contract shared with the HIP kernels: fp32, IEEE add/mul/div/sqrt only, no FMA contraction
 * (-ffp-contract=off on both sides), own polynomial sin/cos/exp/asin/log, fixed evaluation order.  The
 * HIP path is therefore expected to match this file BIT FOR BIT on per-env outputs.
 *
 * Build: oracle/Makefile -> oracle/_build/libtrifinger_oracle.so   (gcc -O2 -ffp-contract=off)
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <time.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef TF_COUNT_FLOPS   /* developer build (make -C oracle flops): this file as C++, every `float` a wrapper that counts its operations */
#include "tf_flops.h"
#endif
#include "../include/trifinger.h"
#include "../include/trifinger_default_caps.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ */
/* deterministic elementary functions (coefficients: Cephes single-precision minimax polynomials)    */
/* ------------------------------------------------------------------------------------------------ */
#ifdef TF_COUNT_FLOPS
#define FMA(a, b, c) cf_fma((a), (b), (c))
#else
#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#endif

/* min / max / clamp with the semantics of the GPU's v_min_f32 / v_max_f32 / v_med3_f32 on non-NaN inputs:
 * a total order in which -0 < +0.  They differ from (a < b) ? a : b only when both operands are zeros of
 * opposite sign: min returns -0 (bitwise OR), max returns +0 (bitwise AND). */
static inline uint32_t f2u_(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f_(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float f_min(float a, float b) {
    float m = (a < b) ? a : b;
    return (a == b) ? u2f_(f2u_(a) | f2u_(b)) : m;
}
static inline float f_max(float a, float b) {
    float m = (a > b) ? a : b;
    return (a == b) ? u2f_(f2u_(a) & f2u_(b)) : m;
}
static inline float f_clamp(float x, float lo, float hi) { return f_min(f_max(x, lo), hi); }   /* median for lo <= hi */
static inline float f_abs(float a) { return u2f_(f2u_(a) & 0x7fffffffu); }   /* |a|: clears the sign bit (also of -0) */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static void tf_sincos(float x, float* s_out, float* c_out) {
    /* Cody-Waite reduction by pi/2, |x| < ~1e4 */
    float k = rintf(x * 0.63661977236758134f);
    int n = (int)k;
    float r = FMA(-k, 1.5703125f, x);
    r = FMA(-k, 4.837512969970703125e-4f, r);
    r = FMA(-k, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = FMA(FMA(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    ps = FMA(ps * z, r, r);
    float pc = FMA(FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    pc = FMA(pc * z, z, FMA(-0.5f, z, 1.0f));
    switch (n & 3) {
        case 0: *s_out = ps; *c_out = pc; break;
        case 1: *s_out = pc; *c_out = -ps; break;
        case 2: *s_out = -ps; *c_out = -pc; break;
        default: *s_out = -pc; *c_out = ps; break;
    }
}

static float tf_exp(float x) {
    x = f_clamp(x, -87.0f, 88.0f);
    float k = rintf(x * 1.44269504088896341f);
    int n = (int)k;
    float r = FMA(-k, 0.693359375f, x);
    r = FMA(k, 2.12194440e-4f, r);
    float z = r * r;
    float p = FMA(FMA(FMA(FMA(FMA(1.9875691500e-4f, r, 1.3981999507e-3f), r, 8.3334519073e-3f), r, 4.1665795894e-2f), r,
                      1.6666665459e-1f), r, 5.0000001201e-1f);
    float e = FMA(p, z, r) + 1.0f;
    return e * u2f((uint32_t)(n + 127) << 23);
}

static float tf_asin(float x) {
    float a = f_abs(x);
    a = f_min(a, 1.0f);
    int big = a > 0.5f;
    float z, y;
    if (big) {
        z = 0.5f * (1.0f - a);
        y = sqrtf(z);
    } else {
        z = a * a;
        y = a;
    }
    float p = FMA(FMA(FMA(FMA(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                  1.6666752422e-1f);
    p = FMA(p * z, y, y);
    if (big) p = 1.5707963267948966f - (p + p);
    return (x < 0.0f) ? -p : p;
}

/* Deterministic reciprocal / reciprocal square root for positive normal x, used for the physics-internal scalings
 * (1/D of the contact rows, unit normals, 1/det ...): integer seed + 3 Newton steps in FMA arithmetic, ~1 ulp (rcp) and
 * ~2 ulp (rsqrt).  Integer and fused multiply-add operations only, so both sides of the parity tests agree bit for
 * bit, and the GPU issues neither the quarter-rate v_rcp/v_sqrt nor the IEEE division / square-root fix-up sequences
 * (11 and 19 issue slots against 7 and 12).  Quantities that the reference defines (rewards, sampling) keep IEEE
 * division and square root. */
static inline float f_rcp(float x) {
    float r = u2f(0x7EF311C7u - f2u(x));
    r = FMA(r, FMA(-x, r, 1.0f), r);
    r = FMA(r, FMA(-x, r, 1.0f), r);
    r = FMA(r, FMA(-x, r, 1.0f), r);
    return r;
}
/* two Newton steps (relative error ~2.4e-4) for the 1/D of a contact row: 1/D only scales the Gauss-Seidel update of
 * that row, its fixed point does not depend on it */
static inline float f_rcp2(float x) {
    float r = u2f_(0x7EF311C7u - f2u_(x));
    r = FMA(r, FMA(-x, r, 1.0f), r);
    r = FMA(r, FMA(-x, r, 1.0f), r);
    return r;
}
static inline float f_rsqrt(float x) {
    float y = u2f(0x5F375A86u - (f2u(x) >> 1));
    const float h = 0.5f * x;
    y = y * FMA(-h, y * y, 1.5f);
    y = y * FMA(-h, y * y, 1.5f);
    y = y * FMA(-h, y * y, 1.5f);
    return y;
}

static float tf_log(float x) {
    /* x > 0, normal.  Cephes logf. */
    uint32_t u = f2u(x);
    int e = (int)((u >> 23) & 0xff) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
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

/* ------------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; Random123)                                                   */
/* ------------------------------------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; } /* [0,1) 24 bit */

/* RNG stream tags (counter word 2) */
enum { RNG_OBJECT = 0, RNG_GOAL_POS = 1, RNG_GOAL_QUAT = 2, RNG_GOAL_ANGVEL = 3, RNG_ROBOT = 4, RNG_DR = 9 /* and 10 */,
       RNG_OBS_NOISE = 16 /* .. 22, counter = frame count instead of reset count */, RNG_ACT_REPEAT = 24 /* counter = frame count */,
       RNG_ACTION = 32 /* .. 36: fused action source, counter = frame count */ };

static void rng4(uint64_t seed, uint32_t env_gid, uint32_t count, uint32_t tag, float u[4]) {
    uint32_t r[4];
    philox4x32_10(env_gid, count, tag, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    for (int i = 0; i < 4; ++i) u[i] = u01(r[i]);
}

/* two uniforms -> two standard normals (Box-Muller) */
static void box_muller(float ua, float ub, float* n0, float* n1) {
    float r = sqrtf(-2.0f * tf_log(1.0f - ua));
    float s, c;
    tf_sincos(6.2831855f * ub, &s, &c);
    *n0 = r * c;
    *n1 = r * s;
}

/* ------------------------------------------------------------------------------------------------ */
/* quaternion helpers (xyzw) - leibnizgym/utils/torch_utils.py:83-150                                */
/* ------------------------------------------------------------------------------------------------ */
static void quat_mul(const float a[4], const float b[4], float o[4]) {
    /* 8-multiplication form, torch_utils.py:99-111 */
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

static float quat_diff_rad(const float a[4], const float b[4]) {
    /* torch_utils.py:131-150: 2 asin(min(|(a * conj(b)).xyz|, 1)) */
    float bc[4] = {-b[0], -b[1], -b[2], b[3]};
    float m[4];
    quat_mul(a, bc, m);
    float nrm = sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    return 2.0f * tf_asin(f_min(nrm, 1.0f));
}

static float lgsk(float x, float scale) {
    /* rewards.py:20-34 */
    float s = x * scale;
    return 1.0f / (tf_exp(s) + 2.0f + tf_exp(-s));
}

static void quat_to_rot(const float q[4], float R[9]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = FMA(-2.0f, FMA(y, y, z * z), 1.0f); R[1] = 2.0f * FMA(x, y, -(w * z));   R[2] = 2.0f * FMA(x, z, w * y);
    R[3] = 2.0f * FMA(x, y, w * z);            R[4] = FMA(-2.0f, FMA(x, x, z * z), 1.0f); R[5] = 2.0f * FMA(y, z, -(w * x));
    R[6] = 2.0f * FMA(x, z, -(w * y));         R[7] = 2.0f * FMA(y, z, w * x);      R[8] = FMA(-2.0f, FMA(x, x, y * y), 1.0f);
}

/* q <- normalize(q + 0.5 h (w,0) * q) */
static void quat_integrate(float q[4], const float w[3], float h) {
    float hx = 0.5f * h * w[0], hy = 0.5f * h * w[1], hz = 0.5f * h * w[2];
    float x = q[0], y = q[1], z = q[2], s = q[3];
    float nx = x + FMA(hx, s, FMA(hy, z, -(hz * y)));
    float ny = y + FMA(hy, s, FMA(hz, x, -(hx * z)));
    float nz = z + FMA(hz, s, FMA(hx, y, -(hy * x)));
    float ns = s - FMA(hx, x, FMA(hy, y, hz * z));
    float inv = f_rsqrt(FMA(nx, nx, FMA(ny, ny, FMA(nz, nz, ns * ns))));
    q[0] = nx * inv; q[1] = ny * inv; q[2] = nz * inv; q[3] = ns * inv;
}

/* ------------------------------------------------------------------------------------------------ */
/* handle                                                                                            */
/* ------------------------------------------------------------------------------------------------ */
#define MAX_OBS 50
#define MAX_STATES 122

struct TfHandle_ {
    TfConfig cfg;
    TfBuffers buf;
    int bound;
    int ext;                            /* general box or extended domain randomisation in use (the HIP library's EXT kernels) */
    int64_t frame_count;
    float clip_obs, clip_act;           /* fused wrapper clipping; FLT_MAX when off */
    int action_dim, obs_dim, states_dim;
    float act_lo[18], act_hi[18];
    float obs_off[MAX_OBS], obs_inv[MAX_OBS];
    float st_off[MAX_STATES], st_inv[MAX_STATES];
    float kp[9], kd[9], ks[9];
    int timing_on;
    double timed_ms;
    int64_t timed_launches;
    float wall_s[3];                    /* slopes of the boundary profile between its knots */
    float wall_c[3], wall_sn[3];        /* cos and sin of the slope angle of each segment (fingertip - boundary contact) */
};

static char g_err[256] = "";

int tf_api_version(void) { return TF_API_VERSION; }
const char* tf_backend_name(void) { return "oracle-c"; }
const char* tf_last_error_string(void) { return g_err; }
int64_t tf_scratch_floats(int32_t num_envs) { (void)num_envs; return 16; }

int tf_action_dim(int32_t mode) {
    if (mode == TF_CMD_TORQUE || mode == TF_CMD_POSITION) return 9;
    if (mode == TF_CMD_POSITION_IMPEDANCE) return 18;
    return TF_ERR_COMMAND_MODE;
}

void tf_default_model(TfModel* m) {
    memset(m, 0, sizeof(*m));
    m->base_height = 0.29f;                                  /* trifingerpro.urdf:51-55 */
    const double yaw[3] = {0.0, -2.09439510239, -4.18879020479}; /* :461-475 */
    for (int f = 0; f < 3; ++f) {
        m->base_yaw_cos[f] = (float)cos(yaw[f]);
        m->base_yaw_sin[f] = (float)sin(yaw[f]);
    }
    m->base_yaw_cos[0] = 1.0f; m->base_yaw_sin[0] = 0.0f;
    for (int f = 0; f < 3; ++f) {
        m->base_half_yaw_cos[f] = (float)cos(0.5 * yaw[f]);
        m->base_half_yaw_sin[f] = (float)sin(0.5 * yaw[f]);
    }
    m->base_half_yaw_cos[0] = 1.0f; m->base_half_yaw_sin[0] = 0.0f;
    m->j2_origin[0] = 0.01685f; m->j2_origin[1] = 0.0505f; m->j2_origin[2] = 0.0f;      /* :180 */
    m->j3_origin[0] = 0.04922f; m->j3_origin[1] = 0.0f;    m->j3_origin[2] = -0.16f;    /* :187 */
    m->tip_origin[0] = 0.0185f; m->tip_origin[1] = 0.0f;   m->tip_origin[2] = -0.1626f; /* :164 */
    /* upper  (:94-98) */
    m->link_mass[0] = 0.26f;
    m->link_com[0][0] = 0.0f; m->link_com[0][1] = 0.06f; m->link_com[0][2] = 0.0f;
    m->link_inertia[0][0] = 0.000459333333333f; m->link_inertia[0][1] = 6.93333333333e-05f;
    m->link_inertia[0][2] = 0.000459333333333f;
    /* middle (:114-118) */
    m->link_mass[1] = 0.25f;
    m->link_com[1][0] = 0.028f; m->link_com[1][1] = 0.0f; m->link_com[1][2] = -0.08f;
    m->link_inertia[1][0] = 0.000441666666667f; m->link_inertia[1][1] = 0.000441666666667f;
    m->link_inertia[1][2] = 6.66666666667e-05f;
    /* lower (:134-138) merged with the rigidly attached tip (:155-164), in double precision */
    {
        const double ml = 0.021, mt = 0.031;
        const double cl[3] = {0.0, 0.0, -0.06}, ct[3] = {0.0185, 0.0, -0.1626};
        const double Il[3] = {3.5e-05, 3.5e-05, 1.4e-06}, It = 5.16666666667e-07;
        double mm = ml + mt, c[3];
        for (int i = 0; i < 3; ++i) c[i] = (ml * cl[i] + mt * ct[i]) / mm;
        double I[6] = {Il[0] + It, Il[1] + It, Il[2] + It, 0.0, 0.0, 0.0};
        const double* cs[2] = {cl, ct};
        const double ms[2] = {ml, mt};
        for (int b = 0; b < 2; ++b) {
            double d[3] = {cs[b][0] - c[0], cs[b][1] - c[1], cs[b][2] - c[2]};
            double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            I[0] += ms[b] * (d2 - d[0] * d[0]);
            I[1] += ms[b] * (d2 - d[1] * d[1]);
            I[2] += ms[b] * (d2 - d[2] * d[2]);
            I[3] += -ms[b] * d[0] * d[1];
            I[4] += -ms[b] * d[0] * d[2];
            I[5] += -ms[b] * d[1] * d[2];
        }
        m->link_mass[2] = (float)mm;
        for (int i = 0; i < 3; ++i) m->link_com[2][i] = (float)c[i];
        for (int i = 0; i < 6; ++i) m->link_inertia[2][i] = (float)I[i];
    }
    for (int i = 0; i < 3; ++i) { /* trifinger_env.py:156-158 */
        static const float lo[3] = {-0.33f, 0.0f, -2.7f}, hi[3] = {1.0f, 1.57f, 0.0f}, df[3] = {0.0f, 0.9f, -1.7f};
        m->q_lo[i] = lo[i]; m->q_hi[i] = hi[i]; m->q_default[i] = df[i];
    }
    m->qd_max = 10.0f;                    /* trifinger_env.py:151 */
    m->tau_max = 0.36f;                   /* :149 */
    m->link_angular_damping = 0.01f;      /* :866 */
    m->cap_a[0] = 0.0135f; m->cap_a[1] = 0.0f; m->cap_a[2] = 0.0f;
    m->cap_b[0] = 0.0185f; m->cap_b[1] = 0.0f; m->cap_b[2] = -0.1592f; /* tip origin + (0,0,0.0034) */
    m->cap_radius = 0.0102f;
    {   /* shapes of the three links: include/trifinger_default_caps.h (fitted to the collision hulls by tools/fit_link_shapes.py) */
        static const TfLinkShape sh3 = TF_DEFAULT_SHAPE3, sh2 = TF_DEFAULT_SHAPE2, sh1 = TF_DEFAULT_SHAPE1;
        static const TfSphere s3[1] = { TF_DEFAULT_SPH3 }, s2[2] = { TF_DEFAULT_SPH2 };
        m->shape3 = sh3; m->shape2 = sh2; m->shape1 = sh1;
        m->sph3[0] = s3[0]; m->sph2[0] = s2[0]; m->sph2[1] = s2[1];
    }
    m->upper_check_z = 0.17f;             /* base height 0.29 - capsule radius - cube half diagonal - margin */
    m->middle_check_z = 0.075f;
    m->cube_half = 0.0325f;               /* trifinger_env.py:143 */
    m->cube_mass = (float)(291.3 * 0.065 * 0.065 * 0.065);
    m->cube_inertia = (float)(291.3 * 0.065 * 0.065 * 0.065 * 0.065 * 0.065 / 6.0);
    m->cube_linear_damping = 0.0f;
    m->cube_angular_damping = 0.05f;
    /* boundary: inner radius by height (measured from convex_table_boundary/convex_*.obj) */
    m->wall_r[0] = 0.1895f; m->wall_z[0] = 0.032f;   /* knots of the piecewise-linear profile r(z): ring, then the flaring cone */
    m->wall_r[1] = 0.2093f; m->wall_z[1] = 0.06f;
    m->wall_r[2] = 0.2319f; m->wall_z[2] = 0.10f;
    m->wall_r[3] = 0.2741f; m->wall_z[3] = 0.176f;
    m->mu_finger_cube = 1.0f;             /* avg(1.0, 1.0) */
    m->mu_cube_floor = 0.55f;             /* avg(1.0, 0.1) */
    m->mu_tip_floor = 0.55f;
    m->mu_cube_wall = 1.0f;
    m->mu_tip_wall = 1.0f;
    m->mu_finger_finger = 1.0f;
    m->mu_robot = 1.0f; m->mu_object = 1.0f; m->mu_floor = 0.1f; m->mu_stage = 1.0f;   /* trifinger_env.py:364-365,876-878,914-915,934-936 */
    m->restitution_finger = 0.4f;         /* avg(0.8, 0.0) */
    m->restitution_ff = 0.8f;             /* avg(0.8, 0.8) */
    m->bounce_threshold = 0.5f;
    m->contact_margin = 0.04f;
    m->contact_slack = 0.005f;
    m->contact_offset = 0.002f;
    m->erp = 0.2f;
    m->warm_start = 0.9f;
    m->max_depenetration_velocity = 1000.0f;
    m->box = 0;
    m->box_gyroscopic = 1;
    m->box_half[0] = 0.0325f; m->box_half[1] = 0.0325f; m->box_half[2] = 0.0325f;
    m->box_inertia[0] = m->cube_inertia; m->box_inertia[1] = m->cube_inertia; m->box_inertia[2] = m->cube_inertia;
    m->obj_radius_3d = 0.05629165f;       /* CuboidalObject(0.065): reference envs/trifinger/utils.py:122-131 */
    m->obj_max_com_dist = 0.13870835f;
    m->obj_min_height = 0.0325f;
    m->obj_span_min_height = 0.0675f;
    m->obj_span_radius = 0.04370835f;
    m->ff_middle_pairs = 1;                /* API 8: the reference keeps every robot link in one self-colliding group (trifinger_env.py:811-812) */
}

/* The object as a general box: mass, principal moments about the body axes, reference inertia of the scaled solve (the mean
 * of the principal moments), CuboidalObject constants (reference envs/trifinger/utils.py:122-131, ARENA_RADIUS :54). */
void tf_model_set_box(TfModel* m, const float size[3], float density) {
    const double sx = (double)size[0], sy = (double)size[1], sz = (double)size[2];
    const double mass = (double)density * sx * sy * sz;
    const double I[3] = {mass * (sy * sy + sz * sz) / 12.0, mass * (sx * sx + sz * sz) / 12.0, mass * (sx * sx + sy * sy) / 12.0};
    {   /* the default cube (65 mm, 291.3 kg/m^3) asked for as a "box" stays the cube: box = 0, headline kernels, isotropic arithmetic */
        TfModel d;
        tf_default_model(&d);
        if ((float)(0.5 * sx) == d.cube_half && (float)(0.5 * sy) == d.cube_half && (float)(0.5 * sz) == d.cube_half && fabs(mass - (double)d.cube_mass) <= 1e-6 * (double)d.cube_mass) {     /* size and density arrive as floats */
            m->box = 0; m->box_gyroscopic = d.box_gyroscopic;
            for (int i = 0; i < 3; ++i) { m->box_half[i] = d.box_half[i]; m->box_inertia[i] = d.box_inertia[i]; }
            m->cube_half = d.cube_half; m->cube_mass = d.cube_mass; m->cube_inertia = d.cube_inertia;
            m->obj_radius_3d = d.obj_radius_3d; m->obj_max_com_dist = d.obj_max_com_dist; m->obj_min_height = d.obj_min_height;
            m->obj_span_min_height = d.obj_span_min_height; m->obj_span_radius = d.obj_span_radius;
            return;
        }
    }
    m->box = 1;
    m->box_gyroscopic = 1;
    m->box_half[0] = (float)(0.5 * sx); m->box_half[1] = (float)(0.5 * sy); m->box_half[2] = (float)(0.5 * sz);
    for (int i = 0; i < 3; ++i) m->box_inertia[i] = (float)I[i];
    m->cube_mass = (float)mass;
    m->cube_inertia = (float)((I[0] + I[1] + I[2]) / 3.0);
    m->cube_half = (float)(0.5 * sz);
    double max_len = sx > sy ? sx : sy;
    if (sz > max_len) max_len = sz;
    const double radius_3d = max_len * sqrt(3.0) / 2.0;
    m->obj_radius_3d = (float)radius_3d;
    m->obj_max_com_dist = (float)(0.195 - radius_3d);
    m->obj_min_height = (float)(sz / 2.0);
    m->obj_span_min_height = (float)(0.1 - sz / 2.0);
    m->obj_span_radius = (float)(0.1 - radius_3d);
}

/* scale tables: trifinger_env.py:153-213 (limits), :655-710 (concatenation order) */
static void build_tables(struct TfHandle_* h) {
    const TfConfig* c = &h->cfg;
    static const float q_lo[3] = {-0.33f, 0.0f, -2.7f}, q_hi[3] = {1.0f, 1.57f, 0.0f};
    static const float kd[3] = {0.1f, 0.3f, 0.001f}, ks[3] = {0.08f, 0.08f, 0.04f};
    int A = h->action_dim;
    for (int j = 0; j < 9; ++j) {
        h->kp[j] = 10.0f; h->kd[j] = kd[j % 3]; h->ks[j] = ks[j % 3];
    }
    for (int j = 0; j < A; ++j) {
        if (c->command_mode == TF_CMD_TORQUE) { h->act_lo[j] = -0.36f; h->act_hi[j] = 0.36f; }
        else if (j < 9) { h->act_lo[j] = q_lo[j % 3]; h->act_hi[j] = q_hi[j % 3]; }
        else { h->act_lo[j] = 1.0f; h->act_hi[j] = 50.0f; }
    }
    float lo[MAX_STATES], hi[MAX_STATES];
    int k = 0;
    for (int j = 0; j < 9; ++j) { lo[k] = q_lo[j % 3]; hi[k] = q_hi[j % 3]; ++k; }
    for (int j = 0; j < 9; ++j) { lo[k] = -10.0f; hi[k] = 10.0f; ++k; }
    for (int rep = 0; rep < 2; ++rep) {
        lo[k] = -0.3f; hi[k] = 0.3f; ++k; lo[k] = -0.3f; hi[k] = 0.3f; ++k; lo[k] = 0.0f; hi[k] = 0.3f; ++k;
        for (int j = 0; j < 4; ++j) { lo[k] = -1.0f; hi[k] = 1.0f; ++k; }
    }
    for (int j = 0; j < A; ++j) {
        if (c->normalize_action) { lo[k] = -1.0f; hi[k] = 1.0f; }
        else { lo[k] = h->act_lo[j]; hi[k] = h->act_hi[j]; }
        ++k;
    }
    h->obs_dim = k;
    for (int j = 0; j < 6; ++j) { lo[k] = -0.5f; hi[k] = 0.5f; ++k; }
    for (int f = 0; f < 3; ++f) {
        lo[k] = -0.4f; hi[k] = 0.4f; ++k; lo[k] = -0.4f; hi[k] = 0.4f; ++k; lo[k] = 0.0f; hi[k] = 0.5f; ++k;
        for (int j = 0; j < 4; ++j) { lo[k] = -1.0f; hi[k] = 1.0f; ++k; }
        for (int j = 0; j < 6; ++j) { lo[k] = -0.2f; hi[k] = 0.2f; ++k; }
    }
    for (int j = 0; j < 9; ++j) { lo[k] = -0.36f; hi[k] = 0.36f; ++k; }
    for (int j = 0; j < 18; ++j) { lo[k] = -1.0f; hi[k] = 1.0f; ++k; }
    h->states_dim = k;
    for (int j = 0; j < h->states_dim; ++j) {
        float off = (lo[j] + hi[j]) * 0.5f;
        float inv = 1.0f / (hi[j] - lo[j]);
        h->st_off[j] = off; h->st_inv[j] = inv;
        if (j < h->obs_dim) { h->obs_off[j] = off; h->obs_inv[j] = inv; }
    }
}

/* mirrors needs_ext() of the HIP library: the base / stage offsets and per-body friction factors take part in the arithmetic
 * only when a box object or one of those randomisations is configured (otherwise not even a + 0.0 is applied) */
static int needs_ext(const TfConfig* c) {
    if (c->model.box) return 1;
    if (!c->dr_enable) return 0;
    for (int i = 0; i < 3; ++i) if (c->dr_base_pos[i] > 0.0f) return 1;
    for (int i = 0; i < 2; ++i) if (c->dr_stage_pos[i] > 0.0f) return 1;
    if (c->dr_friction_robot[0] != 1.0f || c->dr_friction_robot[1] != 1.0f) return 1;
    if (c->dr_friction_object[0] != 1.0f || c->dr_friction_object[1] != 1.0f) return 1;
    if (c->dr_friction_stage[0] != 1.0f || c->dr_friction_stage[1] != 1.0f) return 1;
    return 0;
}

int tf_create(const TfConfig* cfg, tf_handle* out) {
    if (!cfg || !out) return TF_ERR_INVALID_ARG;
    if (cfg->api_version != TF_API_VERSION || cfg->num_envs <= 0) return TF_ERR_INVALID_ARG;
    if (cfg->num_envs > TF_MAX_ENVS) return TF_ERR_INVALID_ARG;
    if (tf_action_dim(cfg->command_mode) < 0) return TF_ERR_COMMAND_MODE;
    if (cfg->robot_reset_type < 0 || cfg->robot_reset_type > 2) return TF_ERR_ROBOT_RESET;
    if (cfg->object_reset_type < 0 || cfg->object_reset_type > 2) return TF_ERR_OBJECT_RESET;
    int d = cfg->task_difficulty;
    if (!(d == -1 || (d >= 1 && d <= 6))) return TF_ERR_DIFFICULTY;
    if (cfg->finger_reach_norm_p != TF_NORM_INF && (cfg->finger_reach_norm_p < 1 || cfg->finger_reach_norm_p > 16)) return TF_ERR_UNSUPPORTED;
    if (cfg->substeps <= 0 || cfg->solver_iterations <= 0 || cfg->solver_inner <= 0 || cfg->control_decimation <= 0 || !(cfg->dt > 0.0f))
        return TF_ERR_INVALID_ARG;
    /* boundary profile (TfModel.wall_z / wall_r, knots of a piecewise-linear r(z) since API 4 - before that: steps of a staircase): the
     * knots must rise strictly and be finite, or the slopes between them are not defined */
    for (int i = 0; i < 4; ++i) {
        const float z = cfg->model.wall_z[i], r = cfg->model.wall_r[i];
        if (!(z - z == 0.0f) || !(r - r == 0.0f) || !(r > 0.0f)) return TF_ERR_INVALID_ARG;
        if (i > 0 && !(z > cfg->model.wall_z[i - 1])) return TF_ERR_INVALID_ARG;
    }
    struct TfHandle_* h = (struct TfHandle_*)calloc(1, sizeof(*h));
    if (h) { h->clip_obs = 3.402823466e38f; h->clip_act = 3.402823466e38f; }
    if (!h) return TF_ERR_INVALID_ARG;
    h->cfg = *cfg;
    if (h->cfg.global_num_envs <= 0) h->cfg.global_num_envs = cfg->num_envs;
    h->action_dim = tf_action_dim(cfg->command_mode);
    build_tables(h);
    h->ext = needs_ext(&h->cfg);
    for (int i = 0; i < 3; ++i) {
        const double sl = ((double)cfg->model.wall_r[i + 1] - (double)cfg->model.wall_r[i]) / ((double)cfg->model.wall_z[i + 1] - (double)cfg->model.wall_z[i]);
        h->wall_s[i] = (float)sl;
        h->wall_c[i] = (float)(1.0 / sqrt(1.0 + sl * sl));
        h->wall_sn[i] = (float)(sl / sqrt(1.0 + sl * sl));
    }
    *out = h;
    return TF_OK;
}

int tf_destroy(tf_handle h) { free(h); return TF_OK; }

int tf_bind(tf_handle h, const TfBuffers* b) {
    if (!h || !b) return TF_ERR_INVALID_ARG;
    if (!b->state || !b->action_buf || !b->obs || !b->reward || !b->reset_buf || !b->goal_reset_buf ||
        !b->successes || !b->dones || !b->steps || !b->reset_count || !b->info || !b->scratch)
        return TF_ERR_INVALID_ARG;
    if (h->cfg.asymmetric_obs && !b->states) return TF_ERR_INVALID_ARG;
    h->buf = *b;
    h->bound = 1;
    return TF_OK;
}

int tf_set_clipping(tf_handle h, float clip_obs, float clip_actions) {
    if (!h) return TF_ERR_INVALID_ARG;
    h->clip_obs = (clip_obs > 0.0f) ? clip_obs : 3.402823466e38f;
    h->clip_act = (clip_actions > 0.0f) ? clip_actions : 3.402823466e38f;
    return TF_OK;
}
int tf_set_gravity(tf_handle h, const float g[3]) {
    if (!h || !g) return TF_ERR_INVALID_ARG;
    for (int i = 0; i < 3; ++i) h->cfg.gravity[i] = g[i];
    return TF_OK;
}
int64_t tf_frame_count(tf_handle h) { return h ? h->frame_count : -1; }
int tf_set_frame_count(tf_handle h, int64_t f) { if (!h) return TF_ERR_INVALID_ARG; h->frame_count = f; return TF_OK; }
/* the product's two instantiations of the fused step share one arithmetic: the oracle has nothing to select (include/trifinger.h) */
int tf_set_kernel_variant(tf_handle h, int32_t variant) { return (!h || variant < TF_KERNEL_AUTO || variant > TF_KERNEL_WIDE_HELPERS) ? TF_ERR_INVALID_ARG : TF_OK; }
int tf_kernel_variant(tf_handle h) { return h ? TF_KERNEL_NARROW : TF_ERR_INVALID_ARG; }
int tf_kernel_occupancy(tf_handle h) { return h ? 0 : TF_ERR_INVALID_ARG; }

/* ------------------------------------------------------------------------------------------------ */
/* finger kinematics and dynamics, in the finger base frame (world = Rz(yaw) * base + (0,0,H))        */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    float s1, c1, s2, c2, s23, c23;
    float p2[3], p3[3];   /* joint-2 / joint-3 origins */
    float ax[3];          /* axis of joints 2 and 3: R1 * e_x */
    float Minv[6];        /* 00 01 02 11 12 22 */
} FK;

/* link k (1..3) frame -> base frame rotation of u */
static inline void rot_link(const FK* k, int link, const float u[3], float o[3]) {
    float wx = u[0], wy = u[1], wz = u[2];
    if (link >= 2) {
        float ca = (link == 2) ? k->c2 : k->c23, sa = (link == 2) ? k->s2 : k->s23;
        float ty = FMA(ca, u[1], -(sa * u[2]));
        float tz = FMA(sa, u[1], ca * u[2]);
        wy = ty; wz = tz;
    }
    o[0] = FMA(k->c1, wx, k->s1 * wz);
    o[1] = wy;
    o[2] = FMA(k->c1, wz, -(k->s1 * wx));
}
/* base frame -> link k frame */
static inline void rot_link_T(const FK* k, int link, const float v[3], float o[3]) {
    float wx = FMA(k->c1, v[0], -(k->s1 * v[2]));
    float wy = v[1];
    float wz = FMA(k->s1, v[0], k->c1 * v[2]);
    if (link >= 2) {
        float ca = (link == 2) ? k->c2 : k->c23, sa = (link == 2) ? k->s2 : k->s23;
        float ty = FMA(ca, wy, sa * wz);
        float tz = FMA(ca, wz, -(sa * wy));
        wy = ty; wz = tz;
    }
    o[0] = wx; o[1] = wy; o[2] = wz;
}
static inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = FMA(a[1], b[2], -(a[2] * b[1]));
    o[1] = FMA(a[2], b[0], -(a[0] * b[2]));
    o[2] = FMA(a[0], b[1], -(a[1] * b[0]));
}
static inline float dot3(const float a[3], const float b[3]) { return FMA(a[2], b[2], FMA(a[1], b[1], a[0] * b[0])); }
/* symmetric inertia (xx yy zz xy xz yz) times vector */
static inline void sym_mul(const float I[6], const float v[3], float o[3]) {
    o[0] = FMA(I[4], v[2], FMA(I[3], v[1], I[0] * v[0]));
    o[1] = FMA(I[5], v[2], FMA(I[1], v[1], I[3] * v[0]));
    o[2] = FMA(I[2], v[2], FMA(I[5], v[1], I[4] * v[0]));
}

static void fk_setup(const TfModel* m, const float q[3], FK* k) {
    tf_sincos(q[0], &k->s1, &k->c1);
    tf_sincos(q[1], &k->s2, &k->c2);
    tf_sincos(q[1] + q[2], &k->s23, &k->c23);
    k->ax[0] = k->c1; k->ax[1] = 0.0f; k->ax[2] = -k->s1;
    rot_link(k, 1, m->j2_origin, k->p2);
    float t[3];
    rot_link(k, 2, m->j3_origin, t);
    k->p3[0] = k->p2[0] + t[0]; k->p3[1] = k->p2[1] + t[1]; k->p3[2] = k->p2[2] + t[2];
}

/* lever arms of the three joints for a base-frame point P: L_j = a_j x (P - p_j) */
static inline void levers(const FK* k, const float P[3], float L1[3], float L2[3], float L3[3]) {
    L1[0] = P[2]; L1[1] = 0.0f; L1[2] = -P[0];
    float r2[3] = {P[0] - k->p2[0], P[1] - k->p2[1], P[2] - k->p2[2]};
    float r3[3] = {P[0] - k->p3[0], P[1] - k->p3[1], P[2] - k->p3[2]};
    cross3(k->ax, r2, L2);
    cross3(k->ax, r3, L3);
}

/* Joint-space mass matrix M (00 01 02 11 12 22) and bias h = C(q,qd) qd + g(q); grav = gravity vector (base frame).
 * Evaluated in the coordinates of link 1 ("frame A": the base frame turned by joint 1 about y).  There joint 1 is the
 * y axis, joints 2 and 3 are the x axis, links 2 and 3 turn about x by q2 and q2+q3, and every vector of the recursive
 * Newton-Euler pass has structural zeros: w_k = (a_k, w, 0) with a_2 = qd2, a_3 = qd2 + qd3 and dw_k = (0, 0, -w a_k),
 * hence  dw x r + w x (w x r) = (w (2 a r_y - w r_x), -a^2 r_y, -(a^2 + w^2) r_z).  Only the components that reach
 * the three joint torques (n1_y, n2_x, n3_x) are formed.  tests/test_physics_analytic.py checks M against the fp64
 * kinetic energy and h against the Lagrangian derivatives of an independent model. */
static void finger_dynamics(const TfModel* m, const FK* k, const float qd[3], const float grav[3], float M[6],
                            float bias[3]) {
    const float m1 = m->link_mass[0], m2 = m->link_mass[1], m3 = m->link_mass[2];
    const float* I1 = m->link_inertia[0];
    const float* I2 = m->link_inertia[1];
    const float* I3 = m->link_inertia[2];
    const float* p2 = m->j2_origin;               /* joint-2 origin and link-1 COM are constants of frame A */
    const float* c1 = m->link_com[0];
    /* frame-A geometry: Rx(a) v = (v_x, c v_y - s v_z, s v_y + c v_z) */
    float d23[3], b[3], e3[3], e2[3];
    d23[0] = m->j3_origin[0];                     /* joint 2 -> joint 3 */
    d23[1] = FMA(k->c2, m->j3_origin[1], -(k->s2 * m->j3_origin[2]));
    d23[2] = FMA(k->s2, m->j3_origin[1], k->c2 * m->j3_origin[2]);
    b[0] = m->link_com[1][0];                     /* joint 2 -> COM 2 */
    b[1] = FMA(k->c2, m->link_com[1][1], -(k->s2 * m->link_com[1][2]));
    b[2] = FMA(k->s2, m->link_com[1][1], k->c2 * m->link_com[1][2]);
    e3[0] = m->link_com[2][0];                    /* joint 3 -> COM 3 */
    e3[1] = FMA(k->c23, m->link_com[2][1], -(k->s23 * m->link_com[2][2]));
    e3[2] = FMA(k->s23, m->link_com[2][1], k->c23 * m->link_com[2][2]);
    e2[0] = d23[0] + e3[0]; e2[1] = d23[1] + e3[1]; e2[2] = d23[2] + e3[2];     /* joint 2 -> COM 3 */
    const float c2x = p2[0] + b[0], c2z = p2[2] + b[2];                           /* COM 2 (x, z) */
    const float c3x = p2[0] + e2[0], c3z = p2[2] + e2[2];                         /* COM 3 (x, z) */
    /* ---- mass matrix: linear part from the COM lever arms L1 = y x P = (P_z, 0, -P_x), L2/L3 = x x r = (0, -r_z, r_y);
     * angular part from the joint axes seen in the link frames, y -> (0, c, -s), x -> x ---- */
    const float u2I = FMA(k->s2 * k->s2, I2[2], FMA(k->c2 * k->c2, I2[1], ((-2.0f * k->c2) * k->s2) * I2[5]));
    const float u3I = FMA(k->s23 * k->s23, I3[2], FMA(k->c23 * k->c23, I3[1], ((-2.0f * k->c23) * k->s23) * I3[5]));
    const float u2x = FMA(k->c2, I2[3], -(k->s2 * I2[4]));
    const float u3x = FMA(k->c23, I3[3], -(k->s23 * I3[4]));
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
    a0[0] = FMA(k->s1, grav[2], -(k->c1 * grav[0]));
    a0[1] = -grav[1];
    a0[2] = -FMA(k->s1, grav[0], k->c1 * grav[2]);
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
        float wl[3] = {a2, k->c2 * w, -(k->s2 * w)}, dl1 = k->s2 * d, dl2 = k->c2 * d;
        float Iw[3], Id[3], t[3];
        sym_mul(I2, wl, Iw);
        Id[0] = FMA(I2[4], dl2, I2[3] * dl1);
        Id[1] = FMA(I2[5], dl2, I2[1] * dl1);
        Id[2] = FMA(I2[2], dl2, I2[5] * dl1);
        cross3(wl, Iw, t);
        const float n0 = Id[0] + t[0], n1 = Id[1] + t[1], n2 = Id[2] + t[2];
        N2x = n0;
        N2y = FMA(k->c2, n1, -(k->s2 * n2));
    }
    {
        const float d = -(w * a3);
        float wl[3] = {a3, k->c23 * w, -(k->s23 * w)}, dl1 = k->s23 * d, dl2 = k->c23 * d;
        float Iw[3], Id[3], t[3];
        sym_mul(I3, wl, Iw);
        Id[0] = FMA(I3[4], dl2, I3[3] * dl1);
        Id[1] = FMA(I3[5], dl2, I3[1] * dl1);
        Id[2] = FMA(I3[2], dl2, I3[5] * dl1);
        cross3(wl, Iw, t);
        const float n0 = Id[0] + t[0], n1 = Id[1] + t[1], n2 = Id[2] + t[2];
        N3x = n0;
        N3y = FMA(k->c23, n1, -(k->s23 * n2));
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

static void inv3sym(const float M[6], float Mi[6]) {
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
static inline void sym3_mul(const float S[6], const float v[3], float o[3]) { /* S = 00 01 02 11 12 22 */
    o[0] = FMA(S[2], v[2], FMA(S[1], v[1], S[0] * v[0]));
    o[1] = FMA(S[4], v[2], FMA(S[3], v[1], S[1] * v[0]));
    o[2] = FMA(S[5], v[2], FMA(S[4], v[1], S[2] * v[0]));
}

/* ------------------------------------------------------------------------------------------------ */
/* per-env working state                                                                             */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    float q[9], qd[9];
    float cp[3], cq[4], cv[3], cw[3];
    float gp[3], gq[4], gw[3];
    float tau[9];
    float ft[18];      /* accumulated fingertip wrench (world), summed over substeps */
    float dr[TF_NUM_DR]; /* domain-randomisation scale factors: cube mass, cube size, friction, motor torque, link mass, restitution */
    /* warm start of the contact solver (state rows TF_S_LAM_* / TF_S_*_LINK|TYPE|FACE) */
    float lam_fc[3][4];  /* normal impulse, world friction impulse */
    float fc_link[3];
    float lam_tf[3][3];  /* fingertip - floor */
    float lam_tw[3][3];  /* fingertip - boundary wall */
    float lam_cf[4][3];
    float cf_face;
    float lam_cw[4][3];
    float cw_face;
} Env;

/* one cube corner against the floor or the wall.  Rows are always evaluated: an inactive contact has
 * (rows of a corner that is not live are skipped) */
typedef struct {
    int active;
    float r[3];
    float n[2];        /* wall: horizontal inward normal */
    float Dinv[3];
    float bias;
    float lam[3];
} CubeContact;

static void base_to_world(const TfModel* m, int f, const float b[3], float w[3]) {
    float c = m->base_yaw_cos[f], s = m->base_yaw_sin[f];
    w[0] = FMA(c, b[0], -(s * b[1]));
    w[1] = FMA(s, b[0], c * b[1]);
    w[2] = b[2] + m->base_height;
}
static void world_to_base(const TfModel* m, int f, const float w[3], float b[3]) {
    float c = m->base_yaw_cos[f], s = m->base_yaw_sin[f];
    b[0] = FMA(c, w[0], s * w[1]);
    b[1] = FMA(c, w[1], -(s * w[0]));
    b[2] = w[2] - m->base_height;
}
static void dir_world_to_base(const TfModel* m, int f, const float w[3], float b[3]) {
    float c = m->base_yaw_cos[f], s = m->base_yaw_sin[f];
    b[0] = FMA(c, w[0], s * w[1]);
    b[1] = FMA(c, w[1], -(s * w[0]));
    b[2] = w[2];
}
static void dir_base_to_world(const TfModel* m, int f, const float b[3], float w[3]) {
    float c = m->base_yaw_cos[f], s = m->base_yaw_sin[f];
    w[0] = FMA(c, b[0], -(s * b[1]));
    w[1] = FMA(s, b[0], c * b[1]);
    w[2] = b[2];
}
/* o = R v and o = R^T v for a row-major 3x3 */
static inline void mat3_mul(const float R[9], const float v[3], float o[3]) {
    o[0] = FMA(R[2], v[2], FMA(R[1], v[1], R[0] * v[0]));
    o[1] = FMA(R[5], v[2], FMA(R[4], v[1], R[3] * v[0]));
    o[2] = FMA(R[8], v[2], FMA(R[7], v[1], R[6] * v[0]));
}
static inline void mat3T_mul(const float R[9], const float v[3], float o[3]) {
    o[0] = FMA(R[6], v[2], FMA(R[3], v[1], R[0] * v[0]));
    o[1] = FMA(R[7], v[2], FMA(R[4], v[1], R[1] * v[0]));
    o[2] = FMA(R[8], v[2], FMA(R[5], v[1], R[2] * v[0]));
}

static void tangent_basis(const float n[3], float t1[3], float t2[3]) {
    if (f_abs(n[2]) < 0.9f) {
        float inv = f_rsqrt(FMA(n[0], n[0], n[1] * n[1]));
        t1[0] = -n[1] * inv; t1[1] = n[0] * inv; t1[2] = 0.0f;
    } else {
        float inv = f_rsqrt(FMA(n[1], n[1], n[2] * n[2]));
        t1[0] = 0.0f; t1[1] = -n[2] * inv; t1[2] = n[1] * inv;
    }
    cross3(n, t1, t2);
}

/* normal-row bias from gap and approach speed (DESIGN.md "contact rows") */
static float contact_bias(const TfModel* m, float gap, float vn0, float inv_h, float restitution) {
    float b;
    if (gap >= 0.0f) b = gap * inv_h;
    else b = f_max(m->erp * gap * inv_h, -m->max_depenetration_velocity);
    if (restitution > 0.0f && gap < m->contact_offset && vn0 < -m->bounce_threshold) b = f_min(b, restitution * vn0);
    return b;
}

/* A contact slot is LIVE (gets rows) when its gap is inside the broad margin and can close within this substep at the
 * approach speed of the free velocities, plus a slack for what other impulses may add: gap < slack + h max(0, -vn0).
 * Everything a dead slot would do is skipped (the kernel masks those lanes; a wavefront with no live lane for a slot
 * skips its rows altogether). */
static inline int contact_live(const TfModel* m, float gap, float vn0, float h) {
    return (gap < m->contact_margin) && (gap < FMA(h, f_max(-vn0, 0.0f), m->contact_slack));
}

/* base-frame position of a point given in the frame of link 1..3 */
static void link_point(const FK* k, int link, const float local[3], float out[3]) {
    float t[3];
    rot_link(k, link, local, t);
    if (link == 1) { out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; }
    else if (link == 2) { out[0] = k->p2[0] + t[0]; out[1] = k->p2[1] + t[1]; out[2] = k->p2[2] + t[2]; }
    else { out[0] = k->p3[0] + t[0]; out[1] = k->p3[1] + t[1]; out[2] = k->p3[2] + t[2]; }
}

/* Joint-space rows of a contact on link `link` of finger f at base-frame point Pb for the three world directions
 * dirs[d]: J[d] = (L1.d, L2.d, L3.d) with the levers of the joints that do not move the link zeroed, W[d] = M^-1 J[d],
 * Dd[d] = J[d].W[d]. */
static void finger_jac(const TfModel* m, int f, const FK* k, int link, const float Pb[3], float dirs[3][3],
                       float J[3][3], float W[3][3], float Dd[3]) {
    float L1[3], L2[3], L3[3];
    levers(k, Pb, L1, L2, L3);
    if (link < 2) { L2[0] = 0.0f; L2[1] = 0.0f; L2[2] = 0.0f; }
    if (link < 3) { L3[0] = 0.0f; L3[1] = 0.0f; L3[2] = 0.0f; }
    for (int d = 0; d < 3; ++d) {
        float db[3];
        dir_world_to_base(m, f, dirs[d], db);
        J[d][0] = dot3(L1, db); J[d][1] = dot3(L2, db); J[d][2] = dot3(L3, db);
        sym3_mul(k->Minv, J[d], W[d]);
        Dd[d] = dot3(J[d], W[d]);
    }
}

/* g(s) = d . (x - clamp(x)) with x = a + s d: half the derivative of the squared distance between the segment point
 * x(s) and the box [-hc, hc]^3; monotone non-decreasing and piecewise linear in s */
static inline float seg_box_g(const float a[3], const float d[3], float s, const float hc[3]) {
    float e[3];
    for (int i = 0; i < 3; ++i) { float x = FMA(s, d[i], a[i]); e[i] = x - f_clamp(x, -hc[i], hc[i]); }
    return dot3(d, e);
}
/* Closest points between the segment a + s (b - a) and the box prod_i [-hc[i], hc[i]], all in the box frame, EXACT: g changes
 * slope only where a coordinate of x(s) crosses +-hc (at most six breakpoints), so the root of g lies on the straight
 * piece between the last breakpoint with g <= 0 and the first with g > 0 (end points included) and is found by one
 * linear interpolation - no iteration (4 alternating projections, the round-1 method, were off by up to 18 mm when the
 * segment runs nearly parallel to a face; tests/test_contact_lcp_reference.py caught it).  x on the segment, y on the
 * box, unit direction nc from y to x, gap = |x - y| - radius.  A segment point inside the box is pushed out through
 * the nearest face. */
static void seg_box(const float a[3], const float b[3], const float hc[3], float radius, float* gap_out, float x[3], float y[3],
                    float nc[3], float* s_out) {
    float d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const float g0 = seg_box_g(a, d, 0.0f, hc), g1 = seg_box_g(a, d, 1.0f, hc);
    float lo = 0.0f, glo = g0, hi = 1.0f, ghi = g1;
    for (int i = 0; i < 3; ++i) {
        const float ad = f_abs(d[i]);
        const int ok = ad > 1e-9f;
        float inv = ok ? f_rcp(ad) : 0.0f;
        inv = (d[i] < 0.0f) ? -inv : inv;
        for (int side = 0; side < 2; ++side) {
            const float sb = ((side ? hc[i] : -hc[i]) - a[i]) * inv;
            const float gb = seg_box_g(a, d, sb, hc);
            const int valid = ok && sb > 0.0f && sb < 1.0f;
            if (valid && gb <= 0.0f && sb > lo) { lo = sb; glo = gb; }
            if (valid && gb > 0.0f && sb < hi) { hi = sb; ghi = gb; }
        }
    }
    float s = f_clamp(FMA(-glo, (hi - lo) * f_rcp(f_max(ghi - glo, 1e-30f)), lo), lo, hi);
    if (g0 > 0.0f) s = 0.0f;
    if (!(g1 > 0.0f)) s = 1.0f;
    *s_out = s;
    for (int i = 0; i < 3; ++i) { x[i] = FMA(s, d[i], a[i]); y[i] = f_clamp(x[i], -hc[i], hc[i]); }
    float ev[3] = {x[0] - y[0], x[1] - y[1], x[2] - y[2]};
    float dist2 = dot3(ev, ev);
    if (dist2 > 1e-12f) {
        float inv = f_rsqrt(dist2);
        float dist = dist2 * inv;
        nc[0] = ev[0] * inv; nc[1] = ev[1] * inv; nc[2] = ev[2] * inv;
        *gap_out = dist - radius;
    } else {
        int bi = 0;
        float best = f_abs(x[0]) - hc[0];
        for (int i = 1; i < 3; ++i) {
            float p = f_abs(x[i]) - hc[i];
            if (p > best) { best = p; bi = i; }
        }
        nc[0] = 0.0f; nc[1] = 0.0f; nc[2] = 0.0f;
        float sg = (x[bi] < 0.0f) ? -1.0f : 1.0f;
        nc[bi] = sg;
        y[bi] = sg * hc[bi];
        *gap_out = best - radius;
    }
}

/* closest points of two segments p1-q1 and p2-q2 (Ericson, Real-Time Collision Detection 5.1.9; both segments have
 * positive length) */
/* the same for a sphere (centre x in the box frame): the tail of seg_box for a segment of zero length */
static void point_box(const float x[3], const float hc[3], float radius, float* gap_out, float y[3], float nc[3]) {
    for (int i = 0; i < 3; ++i) y[i] = f_clamp(x[i], -hc[i], hc[i]);
    float ev[3] = {x[0] - y[0], x[1] - y[1], x[2] - y[2]};
    float dist2 = dot3(ev, ev);
    if (dist2 > 1e-12f) {
        float inv = f_rsqrt(dist2);
        float dist = dist2 * inv;
        nc[0] = ev[0] * inv; nc[1] = ev[1] * inv; nc[2] = ev[2] * inv;
        *gap_out = dist - radius;
    } else {
        int bi = 0;
        float best = f_abs(x[0]) - hc[0];
        float p1 = f_abs(x[1]) - hc[1];
        if (p1 > best) { best = p1; bi = 1; }
        float p2 = f_abs(x[2]) - hc[2];
        if (p2 > best) { best = p2; bi = 2; }
        float xb = x[bi];
        float sg = (xb < 0.0f) ? -1.0f : 1.0f;
        nc[0] = (bi == 0) ? sg : 0.0f; nc[1] = (bi == 1) ? sg : 0.0f; nc[2] = (bi == 2) ? sg : 0.0f;
        y[bi] = sg * hc[bi];
        *gap_out = best - radius;
    }
}

static void seg_seg(const float p1[3], const float q1[3], const float p2[3], const float q2[3], float c1[3], float c2[3]) {
    float d1[3] = {q1[0] - p1[0], q1[1] - p1[1], q1[2] - p1[2]};
    float d2[3] = {q2[0] - p2[0], q2[1] - p2[1], q2[2] - p2[2]};
    float r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    float a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), c = dot3(d1, r), b = dot3(d1, d2);
    float denom = FMA(a, e, -(b * b));
    float ia = f_rcp(a), ie = f_rcp(e);
    float s = 0.0f;
    if (denom > 1e-12f) s = f_clamp(FMA(b, f, -(c * e)) * f_rcp(denom), 0.0f, 1.0f);
    float t = FMA(b, s, f) * ie;
    if (t < 0.0f) { t = 0.0f; s = f_clamp(-c * ia, 0.0f, 1.0f); }
    else if (t > 1.0f) { t = 1.0f; s = f_clamp((b - c) * ia, 0.0f, 1.0f); }
    for (int i = 0; i < 3; ++i) { c1[i] = FMA(s, d1[i], p1[i]); c2[i] = FMA(t, d2[i], p2[i]); }
}

/* the same with the parameter s of the closest point on the first segment (the cross-section of a link shape depends on it) */
static void seg_seg_s(const float p1[3], const float q1[3], const float p2[3], const float q2[3], float c1[3], float c2[3], float* s_out) {
    float d1[3] = {q1[0] - p1[0], q1[1] - p1[1], q1[2] - p1[2]};
    float d2[3] = {q2[0] - p2[0], q2[1] - p2[1], q2[2] - p2[2]};
    float r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    float a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), c = dot3(d1, r), b = dot3(d1, d2);
    float denom = FMA(a, e, -(b * b));
    float ia = f_rcp(a), ie = f_rcp(e);
    float s = 0.0f;
    if (denom > 1e-12f) s = f_clamp(FMA(b, f, -(c * e)) * f_rcp(denom), 0.0f, 1.0f);
    float t = FMA(b, s, f) * ie;
    if (t < 0.0f) { t = 0.0f; s = f_clamp(-c * ia, 0.0f, 1.0f); }
    else if (t > 1.0f) { t = 1.0f; s = f_clamp((b - c) * ia, 0.0f, 1.0f); }
    for (int i = 0; i < 3; ++i) { c1[i] = FMA(s, d1[i], p1[i]); c2[i] = FMA(t, d2[i], p2[i]); }
    *s_out = s;
}

/* inner radius of the boundary at height z: piecewise-linear profile through the knots (wall_z[i], wall_r[i]); a vertical ring below
 * the first knot, nothing above the last (1e3) */
static float wall_radius_at(const struct TfHandle_* H, float z) {
    const TfModel* m = &H->cfg.model;
    float r = m->wall_r[0];
    if (z > m->wall_z[0]) r = FMA(z - m->wall_z[0], H->wall_s[0], m->wall_r[0]);
    if (z > m->wall_z[1]) r = FMA(z - m->wall_z[1], H->wall_s[1], m->wall_r[1]);
    if (z > m->wall_z[2]) r = FMA(z - m->wall_z[2], H->wall_s[2], m->wall_r[2]);
    if (!(z < m->wall_z[3])) r = 1000.0f;
    return r;
}
/* the same with the tilt of the surface at that height: (c, sn) = (cos, sin) of the slope angle of the profile segment, (1, 0) on the vertical ring;
 * inward surface normal (c n_h, sn), distance of a point at radius rho to the surface (r(z) - rho) c.  Fingertip - boundary contact only. */
static float wall_profile(const struct TfHandle_* H, float z, float* c, float* sn) {
    const TfModel* m = &H->cfg.model;
    float r = m->wall_r[0];
    *c = 1.0f; *sn = 0.0f;
    if (z > m->wall_z[0]) { r = FMA(z - m->wall_z[0], H->wall_s[0], m->wall_r[0]); *c = H->wall_c[0]; *sn = H->wall_sn[0]; }
    if (z > m->wall_z[1]) { r = FMA(z - m->wall_z[1], H->wall_s[1], m->wall_r[1]); *c = H->wall_c[1]; *sn = H->wall_sn[1]; }
    if (z > m->wall_z[2]) { r = FMA(z - m->wall_z[2], H->wall_s[2], m->wall_r[2]); *c = H->wall_c[2]; *sn = H->wall_sn[2]; }
    if (!(z < m->wall_z[3])) r = 1000.0f;
    return r;
}

static void cube_corner(const float R[9], const float hc[3], int k, float sk, int idx, float r[3]) {
    int a = (k + 1) % 3, b = (k + 2) % 3;
    float y[3];
    if (a > b) { int t = a; a = b; b = t; }
    y[k] = sk * hc[k];
    y[a] = (idx & 1) ? hc[a] : -hc[a];
    y[b] = (idx & 2) ? hc[b] : -hc[b];
    mat3_mul(R, y, r);
}

/* ---- general box (TfModel.box): rows with explicit arms in INERTIA-SCALED angular coordinates.  With the principal moments
 * I_k, a reference scalar I_ref and S = R diag(sqrt(I_ref / I_k)) R^T (symmetric), the substitution w = S w^, a^ = S a turns
 * a . (I^-1 a) into |a^|^2 / I_ref, a . w into a^ . w^ and w += I^-1 a dl into w^ += a^ dl / I_ref: every row keeps the
 * isotropic form with inv_I = 1 / I_ref, only its arm is S (r x n) instead of r x n. */
static inline void box_arm(const float S[6], const float r[3], const float n[3], float a[3]) {
    float c[3];
    cross3(r, n, c);
    sym3_mul(S, c, a);
}
static inline float g_vrel(const float n[3], const float a[3], const float v[3], const float w[3]) { return dot3(n, v) + dot3(a, w); }
static inline void g_apply(const float n[3], const float a[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    for (int j = 0; j < 3; ++j) { v[j] = FMA(n[j], s, v[j]); w[j] = FMA(a[j], q, w[j]); }
}
/* S = R diag(s) R^T as 00 01 02 11 12 22 */
static void rot_diag_rot(const float R[9], const float s[3], float S[6]) {
    static const int ij[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
    for (int e = 0; e < 6; ++e) {
        const int i = ij[e][0], j = ij[e][1];
        S[e] = FMA(R[3 * i + 2] * s[2], R[3 * j + 2], FMA(R[3 * i + 1] * s[1], R[3 * j + 1], (R[3 * i] * s[0]) * R[3 * j]));
    }
}
static const float BOX_AXES[3][3] = {{0.0f, 0.0f, 1.0f}, {1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}};   /* rows +z, +x, +y of a floor corner */

/* ---- PGS row kernels (identical arithmetic in the HIP file) ---- */
static inline float solve_normal(float* lam, float Dinv, float vrel, float bias) {
    float ln = f_max(FMA(-Dinv, vrel + bias, *lam), 0.0f);
    float dl = ln - *lam;
    *lam = ln;
    return dl;
}
static inline float solve_tangent(float* lam, float Dinv, float vrel, float lim) {
    float ln = f_clamp(FMA(-Dinv, vrel, *lam), -lim, lim);
    float dl = ln - *lam;
    *lam = ln;
    return dl;
}
/* axis-aligned rows of a cube corner with arm r: direction +z / +x / +y.  *_vrel: relative velocity of the row,
 * *_apply: effect of the impulse dl on the cube */
static inline float cz_vrel(const float r[3], const float v[3], const float w[3]) { return FMA(r[1], w[0], FMA(-r[0], w[1], v[2])); }
static inline void cz_apply(const float r[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[2] = v[2] + s;
    w[0] = FMA(r[1], q, w[0]);
    w[1] = FMA(-r[0], q, w[1]);
}
static inline float cx_vrel(const float r[3], const float v[3], const float w[3]) { return FMA(r[2], w[1], FMA(-r[1], w[2], v[0])); }
static inline void cx_apply(const float r[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = v[0] + s;
    w[1] = FMA(r[2], q, w[1]);
    w[2] = FMA(-r[1], q, w[2]);
}
static inline float cy_vrel(const float r[3], const float v[3], const float w[3]) { return FMA(-r[2], w[0], FMA(r[0], w[2], v[1])); }
static inline void cy_apply(const float r[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[1] = v[1] + s;
    w[0] = FMA(-r[2], q, w[0]);
    w[2] = FMA(r[0], q, w[2]);
}
/* wall rows: inward horizontal normal n = (n0, n1, 0) and tangent t = (-n1, n0, 0) */
static inline void wall_arm_n(const CubeContact* c, float a[3]) {
    const float* r = c->r;
    a[0] = -(r[2] * c->n[1]);
    a[1] = r[2] * c->n[0];
    a[2] = FMA(r[0], c->n[1], -(r[1] * c->n[0]));
}
static inline void wall_arm_t(const CubeContact* c, float b[3]) {
    const float* r = c->r;
    b[0] = -(r[2] * c->n[0]);
    b[1] = -(r[2] * c->n[1]);
    b[2] = FMA(r[0], c->n[0], r[1] * c->n[1]);
}
static inline float wn_vrel(const CubeContact* c, const float a[3], const float v[3], const float w[3]) {
    return FMA(a[2], w[2], FMA(a[1], w[1], FMA(a[0], w[0], FMA(c->n[1], v[1], c->n[0] * v[0]))));
}
static inline void wn_apply(const CubeContact* c, const float a[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = FMA(c->n[0], s, v[0]);
    v[1] = FMA(c->n[1], s, v[1]);
    w[0] = FMA(a[0], q, w[0]); w[1] = FMA(a[1], q, w[1]); w[2] = FMA(a[2], q, w[2]);
}
static inline float wt_vrel(const CubeContact* c, const float b[3], const float v[3], const float w[3]) {
    return FMA(b[2], w[2], FMA(b[1], w[1], FMA(b[0], w[0], FMA(c->n[0], v[1], -(c->n[1] * v[0])))));
}
static inline void wt_apply(const CubeContact* c, const float b[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = FMA(-c->n[1], s, v[0]);
    v[1] = FMA(c->n[0], s, v[1]);
    w[0] = FMA(b[0], q, w[0]); w[1] = FMA(b[1], q, w[1]); w[2] = FMA(b[2], q, w[2]);
}

/* arms of the three rows of a wall corner: r x n, r x t and (box only) r x z, inertia-scaled for a box */
static inline void wall_arms(int box, const float S[6], const CubeContact* c, float a[3], float b[3], float c3[3]) {
    wall_arm_n(c, a);
    wall_arm_t(c, b);
    c3[0] = 0.0f; c3[1] = 0.0f; c3[2] = 0.0f;
    if (box) {
        float t[3];
        sym3_mul(S, a, t); a[0] = t[0]; a[1] = t[1]; a[2] = t[2];
        sym3_mul(S, b, t); b[0] = t[0]; b[1] = t[1]; b[2] = t[2];
        box_arm(S, c->r, BOX_AXES[0], c3);
    }
}

/* fingertip sphere against one feature of the arena: finger-only rows */
typedef struct {
    int active;
    float J[3][3], W[3][3], dir[3][3], Dinv[3], bias, lam[3], arm[3], mu;
} TipContact;

/* What one finger keeps through a substep (the "finger role": in the HIP kernel one wavefront per finger). */
typedef struct {
    FK k;
    float vq[3];                  /* joint velocity being solved */
    float Aw[3], Bw[3], Tw[3];    /* distal capsule end points and tip-link origin, world */
    float Bb[3];                  /* tip sphere centre, base frame */
    /* finger-cube contact: joint-space rows (the contact-space record goes to the cube role) */
    int fc_link;
    float fcJ[3][3], fcW[3][3], fc_arm[3];
    /* fingertip sphere against the floor (tc[0]) and against the boundary wall (tc[1]) */
    TipContact tc[2];
    /* joint / velocity limit rows */
    float vlo[3], vhi[3], lim_dinv[3], lim_lam[3];
} FingerRole;

/* Contact-space record of one finger-cube contact, handed to the cube role (LDS in the HIP kernel): the off-diagonal part of the
 * block-local Delassus matrix K = A + D D^T / m + R R^T / I of the three rows (A = J M^-1 J^T the finger side, D the orthonormal
 * directions - D D^T = 1 -, R the cube arms r x d), the cube-side directions and arms, and the contact-point velocity u of the finger
 * side at the start of the sweep.  With K the three rows of a block are solved from the relative velocities at the INCOMING twist:
 * row 1 is corrected by K01 dl0, row 2 by K02 dl0 + K12 dl1 - the Gauss-Seidel iterate of the row-by-row form, with a dependency
 * chain a third as long (the twist and u updates leave the chain). */
typedef struct {
    int active;
    float K[3];                   /* 01 02 12 */
    float Dinv[3], bias, lam[3];
    float dir[3][3], rxd[3][3];
    float u[3], dl[3];
} FcRecord;

/* 1 + s_a (f_a - 1) + s_b (f_b - 1) with the shares s = mu / (mu_a + mu_b) of the two bodies in the pair's average */
static inline float pair_factor(float mu_a, float fa1, float mu_b, float fb1) {
    const float inv = 1.0f / (mu_a + mu_b);
    return FMA(mu_b * inv, fb1, FMA(mu_a * inv, fa1, 1.0f));
}

/* a point given in the frame of link `lk` of finger f, in the cube frame */
static void to_cube(const TfModel* m, int f, const FK* k, int lk, const float local[3], const float cpr[3], const float R[9], float out[3]) {
    float Pb[3], Pw[3];
    link_point(k, lk, local, Pb);
    base_to_world(m, f, Pb, Pw);
    float dd[3] = {Pw[0] - cpr[0], Pw[1] - cpr[1], Pw[2] - cpr[2]};
    mat3T_mul(R, dd, out);
}

/* finger f against the cube: the candidate shapes of its links in turn (a later one takes over only with a strictly smaller gap).  Out: gap,
 * axis point / sphere centre x and closest box point y (cube frame), unit direction nc from y to x, what lies between x and the shape's surface
 * (radius), link (1..3).  Cube pose relative to the robot base: cpr, R; half extents hc. */
static void finger_cube_candidates(const TfModel* m, int f, const FK* k, const float Aw[3], const float Bw[3], const float cpr[3], const float R[9],
                                   const float hc[3], float cube_top_check, float* gap_o, float x[3], float y[3], float nc[3], float* radius_o, int* link_o) {
    float gap = 0.0f, radius = 0.0f;
    int link = 0;
    /* candidates in order (a later one takes over only with a strictly smaller gap): the distal body (TfLinkShape: tapered rounded
     * box along the fingertip capsule's axis), its housing sphere, the middle link, its two housing spheres, and - for a cube above
     * upper_check_z - the upper link */
    for (int pi = 0; pi < 6; ++pi) {
        /* 0: shape3  1: sph3  2: shape2  3: sph2[1] (joint-3 housing)  then, only for a cube above upper_check_z (they hang at the
         * height of the base):  4: sph2[0] (joint-2 housing)  5: shape1 */
        const int lk = (pi < 2) ? 3 : ((pi < 5) ? 2 : 1);
        const TfLinkShape* sh = (pi == 0) ? &m->shape3 : ((pi == 2) ? &m->shape2 : ((pi == 5) ? &m->shape1 : NULL));
        const TfSphere* sp = (pi == 1) ? &m->sph3[0] : ((pi == 3) ? &m->sph2[1] : ((pi == 4) ? &m->sph2[0] : NULL));
        if (pi >= 4 && !(cube_top_check > m->upper_check_z)) continue;
        if (pi == 2 && !(FMA(f_abs(R[8]), hc[2], FMA(f_abs(R[7]), hc[1], FMA(f_abs(R[6]), hc[0], cpr[2]))) > m->middle_check_z))
            continue;                    /* the middle link stays >= 0.12 m above the floor: only an object that reaches up there */
        float gx[3], gy[3], gn[3], gg, rad;
        if (sh) {
            float a[3], b[3], D, spar;
            if (pi == 0) {               /* the axis end points of the distal body are the fingertip capsule's */
                float da[3] = {Aw[0] - cpr[0], Aw[1] - cpr[1], Aw[2] - cpr[2]};
                float db[3] = {Bw[0] - cpr[0], Bw[1] - cpr[1], Bw[2] - cpr[2]};
                mat3T_mul(R, da, a);
                mat3T_mul(R, db, b);
            } else {
                to_cube(m, f, k, lk, sh->a, cpr, R, a);
                to_cube(m, f, k, lk, sh->b, cpr, R, b);
            }
            seg_box(a, b, hc, 0.0f, &D, gx, gy, gn, &spar);
            float uw[3], ub[3], ul[3];
            mat3_mul(R, gn, uw);
            dir_world_to_base(m, f, uw, ub);
            rot_link_T(k, lk, ub, ul);
            const float u1 = -ul[0], u2 = (lk == 1) ? -ul[2] : -ul[1];
            const float rho = FMA(spar, sh->rho[1] - sh->rho[0], sh->rho[0]);
            const float h1 = FMA(spar, sh->w1[1] - sh->w1[0], sh->w1[0]) - rho, h2 = FMA(spar, sh->w2[1] - sh->w2[0], sh->w2[0]) - rho;
            const float o1 = FMA(spar, sh->o1[1] - sh->o1[0], sh->o1[0]), o2 = FMA(spar, sh->o2[1] - sh->o2[0], sh->o2[0]);
            const float ext = FMA(o2, u2, FMA(o1, u1, FMA(h2, f_abs(u2), FMA(h1, f_abs(u1), rho))));
            gg = D - ext;
            rad = ext;
        } else {
            to_cube(m, f, k, lk, sp->c, cpr, R, gx);
            point_box(gx, hc, sp->radius, &gg, gy, gn);
            rad = sp->radius;
        }
        if (link == 0 || gg < gap) {
            link = lk; gap = gg; radius = rad;
            for (int i = 0; i < 3; ++i) { x[i] = gx[i]; y[i] = gy[i]; nc[i] = gn[i]; }
        }
    }
    *gap_o = gap; *radius_o = radius; *link_o = link;
}

/* One solver substep of length h for one env.  Phases and roles: F1 free motion of each finger,
 * C1 cube free motion and corner contacts, FF finger-finger pre-pass, F2 finger contact generation, then the sweeps. */
static void substep(const struct TfHandle_* H, Env* e, float h) {
    const TfConfig* cfg = &H->cfg;
    const TfModel* m = &cfg->model;
    const float inv_h = 1.0f / h;
    /* per-env cube and friction parameters: nominal values times the domain-randomisation factors (1.0 when off) */
    const float cube_mass = m->cube_mass * e->dr[0];
    const float cube_inertia = m->cube_inertia * e->dr[0] * e->dr[1] * e->dr[1];
    const float inv_m = 1.0f / cube_mass, inv_I = 1.0f / cube_inertia;
    /* per-body friction factors: a pair's coefficient is scaled by 1 + s_a (f_a - 1) + s_b (f_b - 1), s = the bodies' shares */
    const float fr1 = e->dr[TF_DR_FRICTION_ROBOT] - 1.0f, fo1 = e->dr[TF_DR_FRICTION_OBJECT] - 1.0f, fs1 = e->dr[TF_DR_FRICTION_STAGE] - 1.0f;
    const int ext = H->ext;
    const float mu_fc = (m->mu_finger_cube * e->dr[2]) * (ext ? pair_factor(m->mu_robot, fr1, m->mu_object, fo1) : 1.0f);
    const float mu_tf = (m->mu_tip_floor * e->dr[2]) * (ext ? pair_factor(m->mu_robot, fr1, m->mu_floor, fs1) : 1.0f);
    const float mu_tw = (m->mu_tip_wall * e->dr[2]) * (ext ? pair_factor(m->mu_robot, fr1, m->mu_stage, fs1) : 1.0f);
    const float mu_cf = (m->mu_cube_floor * e->dr[2]) * (ext ? pair_factor(m->mu_object, fo1, m->mu_floor, fs1) : 1.0f);
    const float mu_cw = (m->mu_cube_wall * e->dr[2]) * (ext ? pair_factor(m->mu_object, fo1, m->mu_stage, fs1) : 1.0f);
    /* robot base offset: the finger roles work in the robot frame = world - offset (the cube position is shifted instead of
     * every kinematic transform); stage offset: centre of the boundary */
    const float* boff = &e->dr[TF_DR_BASE_POS];
    const float* soff = &e->dr[TF_DR_STAGE_POS];
    const float cpr[3] = {ext ? e->cp[0] - boff[0] : e->cp[0], ext ? e->cp[1] - boff[1] : e->cp[1], ext ? e->cp[2] - boff[2] : e->cp[2]};
    const float rest_f = m->restitution_finger * e->dr[5], rest_ff = m->restitution_ff * e->dr[5];
    const float ws = m->warm_start;
    const int box = m->box;
    FingerRole fr[3];
    FcRecord rec[3];
    float v[3], w[3];            /* cube velocities being solved  */
    /* ---- F1: free motion of the fingers ---- */
    for (int f = 0; f < 3; ++f) {
        FingerRole* g = &fr[f];
        float M[6], bias[3], rhs[3], acc[3];
        fk_setup(m, &e->q[3 * f], &g->k);
        finger_dynamics(m, &g->k, &e->qd[3 * f], cfg->gravity, M, bias);
        for (int j = 0; j < 6; ++j) M[j] = M[j] * e->dr[4];       /* link-mass factor: masses and inertias scale together */
        for (int j = 0; j < 3; ++j) bias[j] = bias[j] * e->dr[4];
        inv3sym(M, g->k.Minv);
        for (int j = 0; j < 3; ++j) rhs[j] = e->tau[3 * f + j] - bias[j];
        sym3_mul(g->k.Minv, rhs, acc);
        float damp = 1.0f - h * m->link_angular_damping;
        for (int j = 0; j < 3; ++j) g->vq[j] = FMA(h, acc[j], e->qd[3 * f + j]) * damp;
        float Ab[3], To[3];
        link_point(&g->k, 3, m->cap_a, Ab);
        link_point(&g->k, 3, m->cap_b, g->Bb);
        link_point(&g->k, 3, m->tip_origin, To);
        base_to_world(m, f, Ab, g->Aw);
        base_to_world(m, f, g->Bb, g->Bw);
        base_to_world(m, f, To, g->Tw);
    }
    /* ---- C1: free motion of the cube, corner contacts against the arena ---- */
    float R[9];
    quat_to_rot(e->cq, R);
    float hc[3], S[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < 3; ++k) hc[k] = (box ? m->box_half[k] : m->cube_half) * e->dr[1];
    {
        float wf[3] = {e->cw[0], e->cw[1], e->cw[2]};
        if (box && m->box_gyroscopic) {      /* Euler's equations in the body frame, explicit: I dw/dt = -w x (I w) */
            float wb[3], Iw[3], tq[3];
            mat3T_mul(R, wf, wb);
            for (int k = 0; k < 3; ++k) Iw[k] = m->box_inertia[k] * wb[k];
            cross3(Iw, wb, tq);
            for (int k = 0; k < 3; ++k) wb[k] = FMA(h, tq[k] / m->box_inertia[k], wb[k]);
            mat3_mul(R, wb, wf);
        }
        float dl = 1.0f - h * m->cube_linear_damping, da = 1.0f - h * m->cube_angular_damping;
        for (int i = 0; i < 3; ++i) {
            v[i] = FMA(h, cfg->gravity[i], e->cv[i]) * dl;
            w[i] = wf[i] * da;
        }
        if (box) {                           /* from here to the integration `w` is the inertia-scaled w^ = S^-1 w */
            float sc[3], si[3], Sinv[6], wh[3];
            for (int k = 0; k < 3; ++k) { sc[k] = sqrtf(m->cube_inertia / m->box_inertia[k]); si[k] = 1.0f / sc[k]; }
            rot_diag_rot(R, sc, S);
            rot_diag_rot(R, si, Sinv);
            sym3_mul(Sinv, w, wh);
            for (int i = 0; i < 3; ++i) w[i] = wh[i];
        }
    }
    CubeContact cf[4], cwl[4];
    float cf_face, cw_face;
    {   /* cube vs floor: the four corners of the face that points down most */
        int k = 0;
        float down[3] = {f_abs(R[6]), f_abs(R[7]), f_abs(R[8])};
        if (box) for (int i = 0; i < 3; ++i) down[i] = down[i] * hc[i];     /* the four lowest corners of a box */
        float best = down[0];
        if (down[1] > best) { best = down[1]; k = 1; }
        if (down[2] > best) { best = down[2]; k = 2; }
        float sk = (R[6 + k] > 0.0f) ? -1.0f : 1.0f;
        cf_face = (float)(2 * k + 1 + ((sk > 0.0f) ? 1 : 0));
        const float keep = (cf_face == e->cf_face) ? ws : 0.0f;
        for (int i = 0; i < 4; ++i) {
            CubeContact* c = &cf[i];
            memset(c, 0, sizeof(*c));
            cube_corner(R, hc, k, sk, i, c->r);
            float gap = e->cp[2] + c->r[2];
            float arm[3][3], vn0;
            if (box) {
                for (int d = 0; d < 3; ++d) box_arm(S, c->r, BOX_AXES[d], arm[d]);
                vn0 = g_vrel(BOX_AXES[0], arm[0], v, w);
            } else vn0 = cz_vrel(c->r, v, w);
            if (contact_live(m, gap, vn0, h)) {
                const float* r = c->r;
                c->active = 1;
                if (box) {
                    for (int d = 0; d < 3; ++d) c->Dinv[d] = f_rcp2(FMA(dot3(arm[d], arm[d]), inv_I, inv_m));
                } else {
                    c->Dinv[0] = f_rcp2(FMA(FMA(r[0], r[0], r[1] * r[1]), inv_I, inv_m));   /* +z */
                    c->Dinv[1] = f_rcp2(FMA(FMA(r[2], r[2], r[1] * r[1]), inv_I, inv_m));   /* +x */
                    c->Dinv[2] = f_rcp2(FMA(FMA(r[2], r[2], r[0] * r[0]), inv_I, inv_m));   /* +y */
                }
                c->bias = contact_bias(m, gap, vn0, inv_h, 0.0f);
                for (int d = 0; d < 3; ++d) c->lam[d] = e->lam_cf[i][d] * keep;
            }
        }
    }
    {   /* cube vs boundary wall: the four corners of the face that points outward most */
        const float cx_ = ext ? e->cp[0] - soff[0] : e->cp[0], cy_ = ext ? e->cp[1] - soff[1] : e->cp[1];      /* relative to the stage centre */
        float rc2 = FMA(cx_, cx_, cy_ * cy_);
        float irc = f_rsqrt(f_max(rc2, 1e-24f));
        float rho_c = rc2 * irc;
        int any = rho_c > 1e-6f;
        float dx = 0.0f, dy = 0.0f;
        if (any) { dx = cx_ * irc; dy = cy_ * irc; }
        float pr[3];
        for (int i = 0; i < 3; ++i) pr[i] = FMA(R[i], dx, R[3 + i] * dy);
        int k = 0;
        float out[3] = {f_abs(pr[0]), f_abs(pr[1]), f_abs(pr[2])};
        if (box) for (int i = 0; i < 3; ++i) out[i] = out[i] * hc[i];
        float best = out[0];
        if (out[1] > best) { best = out[1]; k = 1; }
        if (out[2] > best) { best = out[2]; k = 2; }
        float sk = (pr[k] < 0.0f) ? -1.0f : 1.0f;
        /* slot order of the four corners: the lower pair of the face first (the heavy axis - the one whose corner offset has the larger
         * vertical component - picks the pair, the other axis the order inside it); part of the feature the warm start is keyed by */
        const int a_ = (k == 0) ? 1 : 0, b_ = (k == 2) ? 1 : 2;
        const float wa_ = hc[a_] * R[6 + a_], wb_ = hc[b_] * R[6 + b_];
        const int heavy_b = f_abs(wb_) > f_abs(wa_);
        const int lowh = ((heavy_b ? wb_ : wa_) < 0.0f) ? 1 : 0;
        /* inside the lower pair the corner nearer to the boundary comes first, decided anew in every substep; the warm-start rows follow
         * their corner when the order changes.  Feature the warm start is keyed by: face + 8 x pair; stored with the order (+ 32 x order) */
        const float face_pair = (float)(2 * k + 1 + ((sk > 0.0f) ? 1 : 0) + 8 * ((heavy_b ? 2 : 0) + lowh));
        int order, swap01;
        float keep;
        {
            const float prev_o = (e->cw_face >= 32.0f) ? 1.0f : 0.0f;
            float gl[2];
            for (int lb = 0; lb < 2; ++lb) {
                float rr[3];
                cube_corner(R, hc, k, sk, heavy_b ? (lb | (lowh << 1)) : (lowh | (lb << 1)), rr);
                const float qx = cx_ + rr[0], qy = cy_ + rr[1];
                const float q2 = FMA(qx, qx, qy * qy);
                gl[lb] = wall_radius_at(H, e->cp[2] + rr[2]) - q2 * f_rsqrt(f_max(q2, 1e-24f));
            }
            order = (gl[1] < gl[0]) ? 1 : 0;
            swap01 = order ^ (int)prev_o;
            keep = (FMA(-32.0f, prev_o, e->cw_face) == face_pair) ? ws : 0.0f;
        }
        cw_face = FMA(32.0f, (float)order, face_pair);
        for (int i = 0; i < 4; ++i) {
            CubeContact* c = &cwl[i];
            memset(c, 0, sizeof(*c));
            const int hbit = (i >> 1) ^ lowh, lbit = (i < 2) ? ((i & 1) ^ order) : (i & 1);
            cube_corner(R, hc, k, sk, heavy_b ? (lbit | (hbit << 1)) : (hbit | (lbit << 1)), c->r);
            float px = cx_ + c->r[0], py = cy_ + c->r[1], pz = e->cp[2] + c->r[2];
            float rho2 = FMA(px, px, py * py);
            float inv = f_rsqrt(f_max(rho2, 1e-24f));
            float rho = rho2 * inv;
            float gap = wall_radius_at(H, pz) - rho;
            if (!(any && gap < m->contact_margin && rho > 1e-6f)) continue;
            c->n[0] = -px * inv; c->n[1] = -py * inv;
            {
                const float* r = c->r;
                float a[3], b[3], c3[3];
                wall_arms(box, S, c, a, b, c3);
                if (!contact_live(m, gap, wn_vrel(c, a, v, w), h)) { c->n[0] = 0.0f; c->n[1] = 0.0f; continue; }
                c->active = 1;
                c->Dinv[0] = f_rcp2(FMA(dot3(a, a), inv_I, inv_m));
                c->Dinv[1] = f_rcp2(FMA(dot3(b, b), inv_I, inv_m));
                c->Dinv[2] = box ? f_rcp2(FMA(dot3(c3, c3), inv_I, inv_m)) : f_rcp2(FMA(FMA(r[0], r[0], r[1] * r[1]), inv_I, inv_m));
                float vn0 = wn_vrel(c, a, v, w);
                c->bias = contact_bias(m, gap, vn0, inv_h, 0.0f);
                for (int d = 0; d < 3; ++d) c->lam[d] = e->lam_cw[(i < 2) ? (i ^ swap01) : i][d] * keep;
            }
        }
    }
    /* ---- FF: finger-finger contacts (distal capsules), frictionless, resolved before the sweeps on the free velocities:
     * the pairs (0,1), (1,2), (2,0) in turn, one normal row each (a single row is solved exactly by one projection);
     * the finger contacts below measure their approach speeds on the velocities before this pass ---- */
    float vq_ff[3][3];
    for (int f = 0; f < 3; ++f) for (int j = 0; j < 3; ++j) vq_ff[f][j] = fr[f].vq[j];
    for (int p = 0; p < 3; ++p) {
        const int fa = p, fb = (p + 1) % 3;
        float Pa[3], Pb[3];
        seg_seg(fr[fa].Aw, fr[fa].Bw, fr[fb].Aw, fr[fb].Bw, Pa, Pb);
        float dv[3] = {Pa[0] - Pb[0], Pa[1] - Pb[1], Pa[2] - Pb[2]};
        float dist2 = dot3(dv, dv);
        if (!(dist2 > 1e-12f)) continue;
        float inv = f_rsqrt(dist2);
        float dist = dist2 * inv;
        float gap = dist - 2.0f * m->cap_radius;
        if (!(gap < m->contact_margin)) continue;
        float n[3] = {dv[0] * inv, dv[1] * inv, dv[2] * inv};          /* from finger b to finger a */
        float Ja[3], Wa[3], Jb[3], Wb[3];
        for (int side = 0; side < 2; ++side) {
            const int ff = side ? fb : fa;
            const FK* k = &fr[ff].k;
            float C[3], Cb[3], L1[3], L2[3], L3[3], nb[3];
            for (int i = 0; i < 3; ++i) C[i] = side ? FMA(m->cap_radius, n[i], Pb[i]) : FMA(-m->cap_radius, n[i], Pa[i]);
            world_to_base(m, ff, C, Cb);
            levers(k, Cb, L1, L2, L3);
            dir_world_to_base(m, ff, n, nb);
            float* J = side ? Jb : Ja;
            float* W = side ? Wb : Wa;
            J[0] = dot3(L1, nb); J[1] = dot3(L2, nb); J[2] = dot3(L3, nb);
            sym3_mul(k->Minv, J, W);
        }
        float* va = vq_ff[fa];
        float* vb = vq_ff[fb];
        float vn0 = dot3(Ja, va) - dot3(Jb, vb);
        if (!contact_live(m, gap, vn0, h)) continue;
        float bias = contact_bias(m, gap, vn0, inv_h, rest_ff);
        float lam = f_max(-(vn0 + bias) * f_rcp2(dot3(Ja, Wa) + dot3(Jb, Wb)), 0.0f);
        for (int j = 0; j < 3; ++j) { va[j] = FMA(Wa[j], lam, va[j]); vb[j] = FMA(-Wb[j], lam, vb[j]); }
    }
    /* ---- FF, second part (TfModel.ff_middle_pairs, on by default since API 8): the MIDDLE link of finger fm against the distal capsule of each other
     * finger fd - the six ordered pairs, listed as the three with fd = fm + 2 - (0;2) (1;0) (2;1) - and then the three with fd = fm + 1 - (0;1) (1;2)
     * (2;0).  API 8: every one of these rows is solved on the FREE velocities (those
     * before any finger-finger row - a Jacobi step: the rows do not see each other nor the distal pairs), and its two velocity changes d = W lambda
     * (a product rounded on its own) are ADDED to the velocities the distal pairs left, in the order the pairs are listed, a zero component skipped.
     * A row then depends on nothing but what the fingers publish after their free motion, so the three finger wavefronts of the kernels can build
     * the rows of their own middle link side by side and hand the other finger's share over (csrc/tf_roles.h); visited in turn by the cube wavefront -
     * the other placement - the same lines give the same bits; and because the first group is complete before the second begins, a kernel may leave
     * the first group to the cube wavefront and build the second on the finger wavefronts (the 128-register kernels do).
     * The reference leaves every robot link in one collision group with self-collision on (trifinger_env.py:811-812).  The middle link is
     * its finger-cube shape (shape2: tapered rounded box about the axis a -> b of the middle frame), the distal link the fingertip capsule
     * as in the distal pairs.  The cube role of the kernels sees of a finger what it publishes (p2, p3, sin / cos of joint 1, M^-1): the
     * middle frame is rebuilt from those - e_x = (c1, 0, -s1) is the axis of joints 2 and 3, g = (p3 - p2) - j3_x e_x = j3_y e_y + j3_z e_z
     * and e_x x g = j3_y e_z - j3_z e_y give e_y and e_z - and this restatement does it the same way. ---- */
    if (m->ff_middle_pairs) {
        const TfLinkShape* sh = &m->shape2;
        const float jx = m->j3_origin[0], jy = m->j3_origin[1], jz = m->j3_origin[2];
        const float inv_j = f_rcp(FMA(jy, jy, jz * jz));
        for (int o_ = 2; o_ >= 1; --o_) for (int fm = 0; fm < 3; ++fm) {
            const FK* km = &fr[fm].k;
            const float ex[3] = {km->c1, 0.0f, -km->s1};
            float g[3], xg[3], ey[3], ez[3], ab[3], bb[3], aw[3], bw[3];
            for (int i = 0; i < 3; ++i) g[i] = FMA(-jx, ex[i], km->p3[i] - km->p2[i]);
            cross3(ex, g, xg);
            for (int i = 0; i < 3; ++i) { ey[i] = FMA(jy, g[i], -(jz * xg[i])) * inv_j; ez[i] = FMA(jz, g[i], jy * xg[i]) * inv_j; }
            for (int i = 0; i < 3; ++i) {
                ab[i] = FMA(sh->a[2], ez[i], FMA(sh->a[1], ey[i], FMA(sh->a[0], ex[i], km->p2[i])));
                bb[i] = FMA(sh->b[2], ez[i], FMA(sh->b[1], ey[i], FMA(sh->b[0], ex[i], km->p2[i])));
            }
            base_to_world(m, fm, ab, aw);
            base_to_world(m, fm, bb, bw);
            {
                const int o = o_;
                const int fd = (fm + o) % 3;
                float Pm[3], Pd[3], sp;
                seg_seg_s(aw, bw, fr[fd].Aw, fr[fd].Bw, Pm, Pd, &sp);
                float dv[3] = {Pd[0] - Pm[0], Pd[1] - Pm[1], Pd[2] - Pm[2]};
                float dist2 = dot3(dv, dv);
                if (!(dist2 > 1e-12f)) continue;
                float inv = f_rsqrt(dist2);
                float dist = dist2 * inv;
                float n[3] = {dv[0] * inv, dv[1] * inv, dv[2] * inv};      /* from the middle link to the distal capsule */
                float nm[3];
                dir_world_to_base(m, fm, n, nm);
                const float u1 = dot3(nm, ex), u2 = dot3(nm, ey);
                const float rho = FMA(sp, sh->rho[1] - sh->rho[0], sh->rho[0]);
                const float h1 = FMA(sp, sh->w1[1] - sh->w1[0], sh->w1[0]) - rho, h2 = FMA(sp, sh->w2[1] - sh->w2[0], sh->w2[0]) - rho;
                const float o1 = FMA(sp, sh->o1[1] - sh->o1[0], sh->o1[0]), o2 = FMA(sp, sh->o2[1] - sh->o2[0], sh->o2[0]);
                const float ext = FMA(o2, u2, FMA(o1, u1, FMA(h2, f_abs(u2), FMA(h1, f_abs(u1), rho))));
                float gap = dist - ext - m->cap_radius;
                if (!(gap < m->contact_margin)) continue;
                float Jd[3], Wd[3], Jm[3], Wm[3];
                {   /* distal side: the point of the capsule surface that faces the middle link */
                    float C[3], Cb[3], L1[3], L2[3], L3[3], nb[3];
                    for (int i = 0; i < 3; ++i) C[i] = FMA(-m->cap_radius, n[i], Pd[i]);
                    world_to_base(m, fd, C, Cb);
                    levers(&fr[fd].k, Cb, L1, L2, L3);
                    dir_world_to_base(m, fd, n, nb);
                    Jd[0] = dot3(L1, nb); Jd[1] = dot3(L2, nb); Jd[2] = dot3(L3, nb);
                    sym3_mul(fr[fd].k.Minv, Jd, Wd);
                }
                {   /* middle side: joints 1 and 2 move it, joint 3 does not */
                    float C[3], Cb[3], L1[3], L2[3], L3[3];
                    for (int i = 0; i < 3; ++i) C[i] = FMA(ext, n[i], Pm[i]);
                    world_to_base(m, fm, C, Cb);
                    levers(km, Cb, L1, L2, L3);
                    Jm[0] = dot3(L1, nm); Jm[1] = dot3(L2, nm); Jm[2] = 0.0f;
                    sym3_mul(km->Minv, Jm, Wm);
                }
                const float* vd = fr[fd].vq;                        /* free velocities: not vq_ff */
                const float* vm = fr[fm].vq;
                float vn0 = dot3(Jd, vd) - dot3(Jm, vm);
                if (!contact_live(m, gap, vn0, h)) continue;
                float bias = contact_bias(m, gap, vn0, inv_h, rest_ff);
                float lam = f_max(-(vn0 + bias) * f_rcp2(dot3(Jd, Wd) + dot3(Jm, Wm)), 0.0f);
                for (int j = 0; j < 3; ++j) {
                    const float dd = Wd[j] * lam, dm = Wm[j] * lam;
                    if (dd != 0.0f) vq_ff[fd][j] = vq_ff[fd][j] + dd;
                    if (dm != 0.0f) vq_ff[fm][j] = vq_ff[fm][j] - dm;
                }
            }
        }
    }
    /* ---- F2: contacts of each finger ---- */
    float cube_top_check = cpr[2];
    for (int f = 0; f < 3; ++f) {
        FingerRole* g = &fr[f];
        const FK* k = &g->k;
        FcRecord* rc_ = &rec[f];
        memset(rc_, 0, sizeof(*rc_));
        /* --- finger vs cube: the shape with the smallest gap holds the contact --- */
        float gap = 0.0f, x[3], y[3], nc[3], radius = 0.0f;
        int link = 0;
        finger_cube_candidates(m, f, k, g->Aw, g->Bw, cpr, R, hc, cube_top_check, &gap, x, y, nc, &radius, &link);
        g->fc_link = 0;
        for (int d = 0; d < 3; ++d) for (int j = 0; j < 3; ++j) { g->fcJ[d][j] = 0.0f; g->fcW[d][j] = 0.0f; }
        g->fc_arm[0] = 0.0f; g->fc_arm[1] = 0.0f; g->fc_arm[2] = 0.0f;
        if (gap < m->contact_margin) {
            float rcv[3], xw[3], Dd[3], dir[3][3], rxd[3][3], J[3][3], W[3][3];
            mat3_mul(R, nc, dir[0]);
            mat3_mul(R, y, rcv);
            mat3_mul(R, x, xw);
            tangent_basis(dir[0], dir[1], dir[2]);
            /* finger-side contact point (world): axis point minus r n */
            float Pw[3] = {FMA(-radius, dir[0][0], cpr[0] + xw[0]), FMA(-radius, dir[0][1], cpr[1] + xw[1]),
                           FMA(-radius, dir[0][2], cpr[2] + xw[2])};
            float Pb[3];
            world_to_base(m, f, Pw, Pb);
            finger_jac(m, f, k, link, Pb, dir, J, W, Dd);
            for (int d = 0; d < 3; ++d) {
                cross3(rcv, dir[d], rxd[d]);
                if (box) { float t[3]; sym3_mul(S, rxd[d], t); rxd[d][0] = t[0]; rxd[d][1] = t[1]; rxd[d][2] = t[2]; }
            }
            float vn0 = dot3(J[0], g->vq) - (dot3(dir[0], v) + dot3(rxd[0], w));
            if (contact_live(m, gap, vn0, h)) {
                rc_->active = 1;
                g->fc_link = link;
                for (int d = 0; d < 3; ++d) for (int j = 0; j < 3; ++j) {
                    g->fcJ[d][j] = J[d][j]; g->fcW[d][j] = W[d][j]; rc_->dir[d][j] = dir[d][j]; rc_->rxd[d][j] = rxd[d][j];
                }
                for (int d = 0; d < 3; ++d) rc_->Dinv[d] = f_rcp2(FMA(dot3(rxd[d], rxd[d]), inv_I, Dd[d] + inv_m));
                rc_->K[0] = FMA(dot3(rxd[0], rxd[1]), inv_I, dot3(J[0], W[1]));
                rc_->K[1] = FMA(dot3(rxd[0], rxd[2]), inv_I, dot3(J[0], W[2]));
                rc_->K[2] = FMA(dot3(rxd[1], rxd[2]), inv_I, dot3(J[1], W[2]));
                if (link == 3) for (int i = 0; i < 3; ++i) g->fc_arm[i] = Pw[i] - g->Tw[i];
                rc_->bias = contact_bias(m, gap, vn0, inv_h, rest_f);
                if ((float)link == e->fc_link[f]) {          /* same link as in the last substep: seed the impulses */
                    const float* pl = e->lam_fc[f];
                    float l0 = pl[0] * ws;
                    float lim = mu_fc * l0;
                    rc_->lam[0] = l0;
                    rc_->lam[1] = f_clamp(dot3(&pl[1], dir[1]) * ws, -lim, lim);
                    rc_->lam[2] = f_clamp(dot3(&pl[1], dir[2]) * ws, -lim, lim);
                }
            }
        }
        /* --- fingertip sphere vs floor (slot 0) and vs boundary wall (slot 1) --- */
        {
            /* fingertip sphere centre in the world (z) and relative to the stage centre (x, y) */
            const float bx = ext ? (g->Bw[0] + boff[0]) - soff[0] : g->Bw[0], by = ext ? (g->Bw[1] + boff[1]) - soff[1] : g->Bw[1], bz = ext ? g->Bw[2] + boff[2] : g->Bw[2];
            float rho2 = FMA(bx, bx, by * by);
            float inv = f_rsqrt(f_max(rho2, 1e-24f));
            float rho = rho2 * inv;
            float wc, wsn;
            const float wgap = (wall_profile(H, bz, &wc, &wsn) - rho) * wc;      /* distance of the sphere centre to the (tilted) surface */
            const float wall_n[3] = {(-bx * inv) * wc, (-by * inv) * wc, wsn};
            for (int t = 0; t < 2; ++t) {
                TipContact* c = &g->tc[t];
                memset(c, 0, sizeof(*c));
                float gp_ = (t == 0) ? (bz - m->cap_radius) : (wgap - m->cap_radius);
                if (t == 1 && !(rho > 1e-6f)) continue;
                if (!(gp_ < m->contact_margin)) continue;
                float dir[3][3] = {{0.0f, 0.0f, 1.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
                if (t == 1) { dir[0][0] = wall_n[0]; dir[0][1] = wall_n[1]; dir[0][2] = wall_n[2]; }
                tangent_basis(dir[0], dir[1], dir[2]);
                float Pw[3] = {FMA(-m->cap_radius, dir[0][0], g->Bw[0]), FMA(-m->cap_radius, dir[0][1], g->Bw[1]),
                               FMA(-m->cap_radius, dir[0][2], g->Bw[2])};
                float Pb[3], Dd[3], J[3][3], W[3][3];
                world_to_base(m, f, Pw, Pb);
                finger_jac(m, f, k, 3, Pb, dir, J, W, Dd);
                float vn0 = dot3(J[0], g->vq);
                if (!contact_live(m, gp_, vn0, h)) continue;
                c->active = 1;
                c->mu = (t == 0) ? mu_tf : mu_tw;
                for (int d = 0; d < 3; ++d) for (int j = 0; j < 3; ++j) { c->J[d][j] = J[d][j]; c->W[d][j] = W[d][j]; c->dir[d][j] = dir[d][j]; }
                for (int d = 0; d < 3; ++d) c->Dinv[d] = f_rcp2(Dd[d]);
                for (int i = 0; i < 3; ++i) c->arm[i] = Pw[i] - g->Tw[i];
                c->bias = contact_bias(m, gp_, vn0, inv_h, rest_f);
                const float* pl = (t == 0) ? e->lam_tf[f] : e->lam_tw[f];      /* zero when the contact was not there */
                float l0 = pl[0] * ws;
                float lim = c->mu * l0;
                c->lam[0] = l0;
                c->lam[1] = f_clamp(pl[1] * ws, -lim, lim);
                c->lam[2] = f_clamp(pl[2] * ws, -lim, lim);
            }
        }
        /* --- joint limit / velocity limit rows --- */
        for (int jj = 0; jj < 3; ++jj) {
            static const int diag[3] = {0, 3, 5};
            g->vlo[jj] = f_clamp((m->q_lo[jj] - e->q[3 * f + jj]) * inv_h, -m->qd_max, m->qd_max);
            g->vhi[jj] = f_clamp((m->q_hi[jj] - e->q[3 * f + jj]) * inv_h, -m->qd_max, m->qd_max);
            g->lim_dinv[jj] = f_rcp(k->Minv[diag[jj]]);
            g->lim_lam[jj] = 0.0f;
        }
        /* --- velocity after the finger-finger pass, seeded impulses applied on the finger side, contact-point velocity --- */
        for (int j = 0; j < 3; ++j) g->vq[j] = vq_ff[f][j];
        if (rc_->active)
            for (int d = 0; d < 3; ++d) for (int j = 0; j < 3; ++j) g->vq[j] = FMA(g->fcW[d][j], rc_->lam[d], g->vq[j]);
        for (int t = 0; t < 2; ++t) {
            if (!g->tc[t].active) continue;
            for (int d = 0; d < 3; ++d) for (int j = 0; j < 3; ++j) g->vq[j] = FMA(g->tc[t].W[d][j], g->tc[t].lam[d], g->vq[j]);
        }
        if (rc_->active) for (int d = 0; d < 3; ++d) rc_->u[d] = dot3(g->fcJ[d], g->vq);
    }
    /* ---- seeded impulses on the cube side: finger contacts, floor corners, wall corners ---- */
    for (int f = 0; f < 3; ++f) {
        const FcRecord* c = &rec[f];
        if (!c->active) continue;
        for (int d = 0; d < 3; ++d) {
            float sc = c->lam[d] * inv_m, q = c->lam[d] * inv_I;
            for (int j = 0; j < 3; ++j) { v[j] = FMA(-c->dir[d][j], sc, v[j]); w[j] = FMA(-c->rxd[d][j], q, w[j]); }
        }
    }
    for (int i = 0; i < 4; ++i) {
        if (!cf[i].active) continue;
        if (box) {
            for (int d = 0; d < 3; ++d) {
                float a[3];
                box_arm(S, cf[i].r, BOX_AXES[d], a);
                g_apply(BOX_AXES[d], a, cf[i].lam[d], inv_m, inv_I, v, w);
            }
            continue;
        }
        cz_apply(cf[i].r, cf[i].lam[0], inv_m, inv_I, v, w);
        cx_apply(cf[i].r, cf[i].lam[1], inv_m, inv_I, v, w);
        cy_apply(cf[i].r, cf[i].lam[2], inv_m, inv_I, v, w);
    }
    for (int i = 0; i < 4; ++i) {
        if (!cwl[i].active) continue;
        float a[3], b[3], c3[3];
        wall_arms(box, S, &cwl[i], a, b, c3);
        wn_apply(&cwl[i], a, cwl[i].lam[0], inv_m, inv_I, v, w);
        wt_apply(&cwl[i], b, cwl[i].lam[1], inv_m, inv_I, v, w);
        if (box) g_apply(BOX_AXES[0], c3, cwl[i].lam[2], inv_m, inv_I, v, w);
        else cz_apply(cwl[i].r, cwl[i].lam[2], inv_m, inv_I, v, w);
    }
    /* ---- projected Gauss-Seidel ---- */
    /* solver_inner > 1: the block of all rows that touch the cube is visited solver_inner times per sweep, the finger-only rows once (on the last pass) */
    for (int it = 0; it < cfg->solver_iterations * cfg->solver_inner; ++it) {
        const int own_rows = ((it + 1) % cfg->solver_inner) == 0;
        /* cube role: finger-cube rows in contact space */
        for (int f = 0; f < 3; ++f) {
            FcRecord* c = &rec[f];
            if (!c->active) continue;
            float vr[3];
            for (int d = 0; d < 3; ++d) vr[d] = c->u[d] - (dot3(c->dir[d], v) + dot3(c->rxd[d], w));      /* all three at the incoming twist */
            c->dl[0] = solve_normal(&c->lam[0], c->Dinv[0], vr[0], c->bias);
            vr[1] = FMA(c->K[0], c->dl[0], vr[1]);
            c->dl[1] = solve_tangent(&c->lam[1], c->Dinv[1], vr[1], mu_fc * c->lam[0]);
            vr[2] = FMA(c->K[2], c->dl[1], FMA(c->K[1], c->dl[0], vr[2]));
            c->dl[2] = solve_tangent(&c->lam[2], c->Dinv[2], vr[2], mu_fc * c->lam[0]);
            for (int d = 0; d < 3; ++d) {
                float sc = c->dl[d] * inv_m, q = c->dl[d] * inv_I;
                for (int j = 0; j < 3; ++j) { v[j] = FMA(-c->dir[d][j], sc, v[j]); w[j] = FMA(-c->rxd[d][j], q, w[j]); }
            }
        }
        /* finger roles: take the impulses of the finger-cube rows, then the finger-only rows, then publish u */
        for (int f = 0; f < 3; ++f) {
            FingerRole* g = &fr[f];
            FcRecord* c = &rec[f];
            float* vf = g->vq;
            if (c->active) for (int d = 0; d < 3; ++d) for (int j = 0; j < 3; ++j) vf[j] = FMA(g->fcW[d][j], c->dl[d], vf[j]);
            for (int t = 0; t < 2 && own_rows; ++t) { /* fingertip - floor, fingertip - wall */
                TipContact* tcn = &g->tc[t];
                if (!tcn->active) continue;
                for (int d = 0; d < 3; ++d) {
                    float vrel = dot3(tcn->J[d], vf);
                    float dl = (d == 0) ? solve_normal(&tcn->lam[0], tcn->Dinv[0], vrel, tcn->bias)
                                        : solve_tangent(&tcn->lam[d], tcn->Dinv[d], vrel, tcn->mu * tcn->lam[0]);
                    for (int j = 0; j < 3; ++j) vf[j] = FMA(tcn->W[d][j], dl, vf[j]);
                }
            }
            for (int jj = 0; jj < 3 && own_rows; ++jj) {   /* joint limits + velocity limit */
                static const int col[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
                static const int diag[3] = {0, 3, 5};
                const float* Mi = g->k.Minv;
                float v0 = FMA(-Mi[diag[jj]], g->lim_lam[jj], vf[jj]);
                float tgt = f_clamp(v0, g->vlo[jj], g->vhi[jj]);
                float lam_new = (tgt - v0) * g->lim_dinv[jj];
                float dl = lam_new - g->lim_lam[jj];
                g->lim_lam[jj] = lam_new;
                vf[0] = FMA(Mi[col[jj][0]], dl, vf[0]);
                vf[1] = FMA(Mi[col[jj][1]], dl, vf[1]);
                vf[2] = FMA(Mi[col[jj][2]], dl, vf[2]);
            }
            if (c->active) for (int d = 0; d < 3; ++d) c->u[d] = dot3(g->fcJ[d], vf);
        }
        /* cube role: corner rows */
        for (int i = 0; i < 4; ++i) {             /* cube - floor: rows +z (normal), +x, +y */
            CubeContact* c = &cf[i];
            if (!c->active) continue;
            if (box) {
                for (int d = 0; d < 3; ++d) {
                    float a[3];
                    box_arm(S, c->r, BOX_AXES[d], a);
                    float vrel = g_vrel(BOX_AXES[d], a, v, w);
                    float dlb = (d == 0) ? solve_normal(&c->lam[0], c->Dinv[0], vrel, c->bias)
                                         : solve_tangent(&c->lam[d], c->Dinv[d], vrel, mu_cf * c->lam[0]);
                    g_apply(BOX_AXES[d], a, dlb, inv_m, inv_I, v, w);
                }
                continue;
            }
            float dl = solve_normal(&c->lam[0], c->Dinv[0], cz_vrel(c->r, v, w), c->bias);
            cz_apply(c->r, dl, inv_m, inv_I, v, w);
            dl = solve_tangent(&c->lam[1], c->Dinv[1], cx_vrel(c->r, v, w), mu_cf * c->lam[0]);
            cx_apply(c->r, dl, inv_m, inv_I, v, w);
            dl = solve_tangent(&c->lam[2], c->Dinv[2], cy_vrel(c->r, v, w), mu_cf * c->lam[0]);
            cy_apply(c->r, dl, inv_m, inv_I, v, w);
        }
        for (int i = 0; i < 4; ++i) {             /* cube - wall: rows n (normal), t, +z */
            CubeContact* c = &cwl[i];
            if (!c->active) continue;
            float a[3], b[3], c3[3];
            wall_arms(box, S, c, a, b, c3);
            float dl = solve_normal(&c->lam[0], c->Dinv[0], wn_vrel(c, a, v, w), c->bias);
            wn_apply(c, a, dl, inv_m, inv_I, v, w);
            dl = solve_tangent(&c->lam[1], c->Dinv[1], wt_vrel(c, b, v, w), mu_cw * c->lam[0]);
            wt_apply(c, b, dl, inv_m, inv_I, v, w);
            if (box) {
                dl = solve_tangent(&c->lam[2], c->Dinv[2], g_vrel(BOX_AXES[0], c3, v, w), mu_cw * c->lam[0]);
                g_apply(BOX_AXES[0], c3, dl, inv_m, inv_I, v, w);
            } else {
                dl = solve_tangent(&c->lam[2], c->Dinv[2], cz_vrel(c->r, v, w), mu_cw * c->lam[0]);
                cz_apply(c->r, dl, inv_m, inv_I, v, w);
            }
        }
    }
    /* ---- impulses kept for the next substep; fingertip wrench sensor: contact impulses / h, world frame, about the
     * tip-link origin (contacts on the distal link and the fingertip only) ---- */
    for (int f = 0; f < 3; ++f) {
        const FingerRole* g = &fr[f];
        const FcRecord* c = &rec[f];
        float ftv[3], Fc[3];
        for (int i = 0; i < 3; ++i) {
            ftv[i] = FMA(c->dir[2][i], c->lam[2], c->dir[1][i] * c->lam[1]);
            Fc[i] = FMA(c->dir[0][i], c->lam[0], ftv[i]) * inv_h;
        }
        e->lam_fc[f][0] = c->lam[0];
        for (int i = 0; i < 3; ++i) e->lam_fc[f][1 + i] = ftv[i];
        e->fc_link[f] = (float)g->fc_link;
        for (int d = 0; d < 3; ++d) { e->lam_tf[f][d] = g->tc[0].lam[d]; e->lam_tw[f][d] = g->tc[1].lam[d]; }
        if (cfg->asymmetric_obs && g->fc_link == 3) {
            float T[3];
            cross3(g->fc_arm, Fc, T);
            for (int i = 0; i < 3; ++i) { e->ft[6 * f + i] += Fc[i]; e->ft[6 * f + 3 + i] += T[i]; }
        }
        for (int t = 0; t < 2; ++t) {
            const TipContact* tcn = &g->tc[t];
            if (!tcn->active || !cfg->asymmetric_obs) continue;
            float F[3], T[3];
            for (int i = 0; i < 3; ++i)
                F[i] = FMA(tcn->dir[2][i], tcn->lam[2], FMA(tcn->dir[1][i], tcn->lam[1], tcn->dir[0][i] * tcn->lam[0])) * inv_h;
            cross3(tcn->arm, F, T);
            for (int i = 0; i < 3; ++i) { e->ft[6 * f + i] += F[i]; e->ft[6 * f + 3 + i] += T[i]; }
        }
    }
    for (int i = 0; i < 4; ++i) for (int d = 0; d < 3; ++d) { e->lam_cf[i][d] = cf[i].lam[d]; e->lam_cw[i][d] = cwl[i].lam[d]; }
    e->cf_face = cf_face;
    {   /* 0 while no corner touches the boundary: the wall-corner rows then carry nothing */
        int any_wall = 0;
        for (int i = 0; i < 4; ++i) any_wall |= cwl[i].active;
        e->cw_face = any_wall ? cw_face : 0.0f;
    }
    /* ---- integrate ---- */
    for (int f = 0; f < 3; ++f) for (int jj = 0; jj < 3; ++jj) {
        const int j = 3 * f + jj;
        e->qd[j] = fr[f].vq[jj];
        e->q[j] = f_clamp(FMA(h, fr[f].vq[jj], e->q[j]), m->q_lo[jj], m->q_hi[jj]);
    }
    if (box) {                               /* back to the world angular velocity: w = S w^ */
        float ww[3];
        sym3_mul(S, w, ww);
        for (int i = 0; i < 3; ++i) w[i] = ww[i];
    }
    for (int i = 0; i < 3; ++i) {
        e->cv[i] = v[i]; e->cw[i] = w[i];
        e->cp[i] = FMA(h, v[i], e->cp[i]);
    }
    quat_integrate(e->cq, e->cw, h);
}

/* the moving goal (goal_movement.rotation) is a free body nothing interacts with: its orientation is integrated with
 * the same substep sequence AFTER observations, rewards and termination of the step have used the pose the step started
 * with (reference trifinger_env.py:500-559: __update_goal_movement_post comes last) */
static void goal_advance(const struct TfHandle_* H, Env* e, int nsub, float h) {
    if (H->cfg.goal_rotation_activate) for (int s = 0; s < nsub; ++s) quat_integrate(e->gq, e->gw, h);
}

/* ------------------------------------------------------------------------------------------------ */
/* SoA <-> Env                                                                                        */
/* ------------------------------------------------------------------------------------------------ */
#define ST(h, row, i) ((h)->buf.state[(size_t)(row) * (size_t)(h)->cfg.num_envs + (size_t)(i)])

static void env_load(const struct TfHandle_* h, int i, Env* e) {
    for (int j = 0; j < 9; ++j) { e->q[j] = ST(h, TF_S_Q + j, i); e->qd[j] = ST(h, TF_S_QD + j, i); e->tau[j] = ST(h, TF_S_TAU + j, i); }
    for (int j = 0; j < 3; ++j) {
        e->cp[j] = ST(h, TF_S_CUBE_P + j, i); e->cv[j] = ST(h, TF_S_CUBE_V + j, i); e->cw[j] = ST(h, TF_S_CUBE_W + j, i);
        e->gp[j] = ST(h, TF_S_GOAL_P + j, i); e->gw[j] = ST(h, TF_S_GOAL_W + j, i);
    }
    for (int j = 0; j < 4; ++j) { e->cq[j] = ST(h, TF_S_CUBE_Q + j, i); e->gq[j] = ST(h, TF_S_GOAL_Q + j, i); }
    for (int j = 0; j < 18; ++j) e->ft[j] = ST(h, TF_S_FT + j, i);
    for (int j = 0; j < TF_NUM_DR; ++j) e->dr[j] = (h->cfg.dr_enable && (h->ext || j < TF_DR_BASE_POS)) ? ST(h, TF_S_DR + j, i) : TF_DR_NEUTRAL(j);   /* rows are read only when the feature is on */
    /* warm-start rows.  Row TF_S_FC_LINK + f is the activity code of finger f: link that held the finger-cube contact + 4 if the
     * fingertip-wall contact pushed; TF_S_CW_FACE is 0 while no corner of the cube touches the boundary.  The rows of an inactive
     * contact are neither read nor written (their content is undefined). */
    for (int f = 0; f < 3; ++f) {
        const int code = (int)ST(h, TF_S_FC_LINK + f, i);
        e->fc_link[f] = (float)(code & 3);
        for (int j = 0; j < 4; ++j) e->lam_fc[f][j] = (code & 3) ? ST(h, TF_S_LAM_FC + 4 * f + j, i) : 0.0f;
        for (int j = 0; j < 3; ++j) {
            e->lam_tf[f][j] = ST(h, TF_S_LAM_TF + 3 * f + j, i);
            e->lam_tw[f][j] = (code & 4) ? ST(h, TF_S_LAM_TW + 3 * f + j, i) : 0.0f;
        }
    }
    e->cf_face = ST(h, TF_S_CF_FACE, i);
    e->cw_face = ST(h, TF_S_CW_FACE, i);
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 3; ++j) {
        e->lam_cf[c][j] = ST(h, TF_S_LAM_CF + 3 * c + j, i);
        e->lam_cw[c][j] = (e->cw_face != 0.0f) ? ST(h, TF_S_LAM_CW + 3 * c + j, i) : 0.0f;
    }
}
static void env_store(const struct TfHandle_* h, int i, const Env* e, int store_ft) {
    for (int j = 0; j < 9; ++j) { ST(h, TF_S_Q + j, i) = e->q[j]; ST(h, TF_S_QD + j, i) = e->qd[j]; ST(h, TF_S_TAU + j, i) = e->tau[j]; }
    for (int j = 0; j < 3; ++j) {
        ST(h, TF_S_CUBE_P + j, i) = e->cp[j]; ST(h, TF_S_CUBE_V + j, i) = e->cv[j]; ST(h, TF_S_CUBE_W + j, i) = e->cw[j];
        ST(h, TF_S_GOAL_P + j, i) = e->gp[j]; ST(h, TF_S_GOAL_W + j, i) = e->gw[j];
    }
    for (int j = 0; j < 4; ++j) { ST(h, TF_S_CUBE_Q + j, i) = e->cq[j]; ST(h, TF_S_GOAL_Q + j, i) = e->gq[j]; }
    if (store_ft) for (int j = 0; j < 18; ++j) ST(h, TF_S_FT + j, i) = e->ft[j];
    if (h->cfg.dr_enable) for (int j = 0; j < (h->ext ? TF_NUM_DR : TF_DR_BASE_POS); ++j) ST(h, TF_S_DR + j, i) = e->dr[j];
    for (int f = 0; f < 3; ++f) {
        const int link = (int)e->fc_link[f], tw_now = e->lam_tw[f][0] > 0.0f;
        ST(h, TF_S_FC_LINK + f, i) = (float)(link + (tw_now ? 4 : 0));
        if (link != 0) for (int j = 0; j < 4; ++j) ST(h, TF_S_LAM_FC + 4 * f + j, i) = e->lam_fc[f][j];
        for (int j = 0; j < 3; ++j) ST(h, TF_S_LAM_TF + 3 * f + j, i) = e->lam_tf[f][j];
        if (tw_now) for (int j = 0; j < 3; ++j) ST(h, TF_S_LAM_TW + 3 * f + j, i) = e->lam_tw[f][j];
    }
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 3; ++j) {
        ST(h, TF_S_LAM_CF + 3 * c + j, i) = e->lam_cf[c][j];
        if (e->cw_face != 0.0f) ST(h, TF_S_LAM_CW + 3 * c + j, i) = e->lam_cw[c][j];
    }
    ST(h, TF_S_CF_FACE, i) = e->cf_face;
    ST(h, TF_S_CW_FACE, i) = e->cw_face;
}

/* ------------------------------------------------------------------------------------------------ */
/* task layer                                                                                        */
/* ------------------------------------------------------------------------------------------------ */
/* sample.py:22-34 */
static void sample_xy(float u_r, float u_t, float r_max, float* x, float* y) {
    float radius = sqrtf(u_r) * r_max;
    float s, c;
    tf_sincos(6.2831855f * u_t, &s, &c);
    *x = radius * c;
    *y = radius * s;
}
/* sample.py:77-84 via torch_utils.py:153-180 with roll = pitch = 0 */
static void sample_yaw_quat(float u, float q[4]) {
    float s, c;
    tf_sincos((6.2831855f * u) * 0.5f, &s, &c);
    q[0] = 0.0f; q[1] = 0.0f; q[2] = s; q[3] = c;
}
/* sample.py:55-65: normalize(randn(4)), eps 1e-12 */
static void normalize_quat(const float n[4], float q[4]) {
    float nrm = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3]);
    float inv = 1.0f / f_max(nrm, 1e-12f);
    for (int i = 0; i < 4; ++i) q[i] = n[i] * inv;
}

#define CUBE_RADIUS_3D 0.05629165f      /* CuboidalObject(0.065).radius_3d,  envs/trifinger/utils.py:122-131 */
#define CUBE_MAX_COM_DIST 0.13870835f   /* ARENA_RADIUS - radius_3d                                            */
#define CUBE_MIN_HEIGHT 0.0325f
#define CUBE_MAX_HEIGHT 0.1f

/* trifinger_env.py:1194-1265; the position relative to the stage centre (raw[0..2]), orientation (raw[3..6]), angular velocity (raw[7..9]) */
static void sample_goal_raw(const struct TfHandle_* h, uint32_t gid, uint32_t count, float raw[10]) {
    const TfConfig* c = &h->cfg;
    int d = c->task_difficulty;
    float u[4];
    rng4(c->seed, gid, count, RNG_GOAL_POS, u);
    float x = 0.0f, y = 0.0f, z;
    float quat[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    if (d == -1 || d == 1 || d == 3 || d == 4 || d == 5) sample_xy(u[0], u[1], h->cfg.model.obj_max_com_dist, &x, &y);
    if (d == -1 || d == 1) z = h->cfg.model.obj_min_height;
    else if (d == 2 || d == 6) z = h->cfg.model.obj_min_height + 0.05f;
    else if (d == 3) z = h->cfg.model.obj_span_min_height * u[2] + h->cfg.model.obj_min_height;           /* (max_height - min_height) in double, then fp32 */
    else z = h->cfg.model.obj_span_radius * u[2] + h->cfg.model.obj_radius_3d;                    /* (max_height - radius_3d) */
    if (d == -1) sample_yaw_quat(u[3], quat);
    if (d == 4 || d == 5 || d == 6) {
        float v[4], n[4];
        rng4(c->seed, gid, count, RNG_GOAL_QUAT, v);
        box_muller(v[0], v[1], &n[0], &n[1]);
        box_muller(v[2], v[3], &n[2], &n[3]);
        normalize_quat(n, quat);
    }
    if (c->goal_rotation_activate) {     /* sample.py:67-75 */
        float v[4], n[4];
        rng4(c->seed, gid, count, RNG_GOAL_ANGVEL, v);
        box_muller(v[0], v[1], &n[0], &n[1]);
        box_muller(v[2], v[3], &n[2], &n[3]);
        float nrm = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        float mag = n[3] * c->goal_rotation_rate_magnitude;
        for (int i = 0; i < 3; ++i) raw[7 + i] = mag * (n[i] / nrm);
    } else {
        raw[7] = 0.0f; raw[8] = 0.0f; raw[9] = 0.0f;
    }
    raw[0] = x; raw[1] = y; raw[2] = z;
    for (int i = 0; i < 4; ++i) raw[3 + i] = quat[i];
}
static void sample_goal(const struct TfHandle_* h, uint32_t gid, uint32_t count, Env* e) {
    float raw[10];
    sample_goal_raw(h, gid, count, raw);
    e->gp[0] = h->ext ? raw[0] + e->dr[TF_DR_STAGE_POS] : raw[0]; e->gp[1] = h->ext ? raw[1] + e->dr[TF_DR_STAGE_POS + 1] : raw[1]; e->gp[2] = raw[2];   /* goals move with the stage */
    for (int i = 0; i < 4; ++i) e->gq[i] = raw[3 + i];
    for (int i = 0; i < 3; ++i) e->gw[i] = raw[7 + i];
}
/* The samples of the env's NEXT reset, formed at the end of the fused step that flags it (include/trifinger.h: TF_S_NEXT_*).  The kernels load them at the
 * reset when the tag matches; the values are what the reset draws itself, so this restatement only has to WRITE the same rows at the same time. */
static void presample_next_reset(const struct TfHandle_* h, int i) {
    const TfConfig* c = &h->cfg;
    const uint32_t gid = (uint32_t)(c->env_id_offset + i), count = h->buf.reset_count[i];
    if (c->object_reset_type == TF_RESET_RANDOM) {
        float u[4], xy[2], q[4];
        rng4(c->seed, gid, count, RNG_OBJECT, u);
        sample_xy(u[0], u[1], c->model.obj_max_com_dist, &xy[0], &xy[1]);
        sample_yaw_quat(u[2], q);
        ST(h, TF_S_NEXT_OBJ + 0, i) = xy[0]; ST(h, TF_S_NEXT_OBJ + 1, i) = xy[1]; ST(h, TF_S_NEXT_OBJ + 2, i) = q[2]; ST(h, TF_S_NEXT_OBJ + 3, i) = q[3];
    }
    float raw[10];
    sample_goal_raw(h, gid, count, raw);
    for (int j = 0; j < 10; ++j) ST(h, TF_S_NEXT_GOAL + j, i) = raw[j];
    const uint32_t tag = count + 1u;
    memcpy(&ST(h, TF_S_NEXT_TAG, i), &tag, sizeof(tag));
}

/* masked _reset_impl then _goal_reset_impl (env_base.py:370-379; trifinger_env.py:373-440).
 * Returns 1 if the env was reset (its action row must be zeroed). */
static int apply_resets(const struct TfHandle_* h, int i, Env* e, int force_all) {
    const TfConfig* c = &h->cfg;
    const TfModel* m = &c->model;
    uint32_t gid = (uint32_t)(c->env_id_offset + i);
    int did_reset = 0;
    if (force_all || h->buf.reset_buf[i]) {
        did_reset = 1;
        uint32_t count = h->buf.reset_count[i];
        h->buf.reset_buf[i] = 0;
        h->buf.steps[i] = 0;
        h->buf.successes[i] = 0;
        for (int j = 0; j < 9; ++j) e->tau[j] = 0.0f;      /* the stored torque (what an action repeat re-applies) */
        memset(e->lam_fc, 0, sizeof(e->lam_fc)); memset(e->lam_tf, 0, sizeof(e->lam_tf)); memset(e->lam_tw, 0, sizeof(e->lam_tw));   /* solver warm start */
        memset(e->lam_cf, 0, sizeof(e->lam_cf)); memset(e->lam_cw, 0, sizeof(e->lam_cw));
        for (int f = 0; f < 3; ++f) e->fc_link[f] = 0.0f;
        e->cf_face = 0.0f; e->cw_face = 0.0f;
        if (c->dr_enable) {     /* build-defined domain randomisation: scale = lo + (hi - lo) u */
            float u[4];
            rng4(c->seed, gid, count, RNG_DR, u);
            e->dr[0] = FMA(c->dr_cube_mass[1] - c->dr_cube_mass[0], u[0], c->dr_cube_mass[0]);
            e->dr[1] = FMA(c->dr_cube_size[1] - c->dr_cube_size[0], u[1], c->dr_cube_size[0]);
            e->dr[2] = FMA(c->dr_friction[1] - c->dr_friction[0], u[2], c->dr_friction[0]);
            e->dr[3] = FMA(c->dr_motor[1] - c->dr_motor[0], u[3], c->dr_motor[0]);
            rng4(c->seed, gid, count, RNG_DR + 1u, u);
            e->dr[4] = FMA(c->dr_link_mass[1] - c->dr_link_mass[0], u[0], c->dr_link_mass[0]);
            e->dr[5] = FMA(c->dr_restitution[1] - c->dr_restitution[0], u[1], c->dr_restitution[0]);
            /* robot base and stage positions: offset = a (2 u - 1) per axis; friction per body */
            if (h->ext) {
            rng4(c->seed, gid, count, RNG_DR + 2u, u);
            for (int k = 0; k < 3; ++k) e->dr[TF_DR_BASE_POS + k] = c->dr_base_pos[k] * (2.0f * u[k] - 1.0f);
            e->dr[TF_DR_STAGE_POS] = c->dr_stage_pos[0] * (2.0f * u[3] - 1.0f);
            rng4(c->seed, gid, count, RNG_DR + 3u, u);
            e->dr[TF_DR_STAGE_POS + 1] = c->dr_stage_pos[1] * (2.0f * u[0] - 1.0f);
            e->dr[TF_DR_FRICTION_ROBOT] = FMA(c->dr_friction_robot[1] - c->dr_friction_robot[0], u[1], c->dr_friction_robot[0]);
            e->dr[TF_DR_FRICTION_OBJECT] = FMA(c->dr_friction_object[1] - c->dr_friction_object[0], u[2], c->dr_friction_object[0]);
            e->dr[TF_DR_FRICTION_STAGE] = FMA(c->dr_friction_stage[1] - c->dr_friction_stage[0], u[3], c->dr_friction_stage[0]);
            }
        }
        if (c->robot_reset_type == TF_RESET_DEFAULT) {
            for (int j = 0; j < 9; ++j) { e->q[j] = m->q_default[j % 3]; e->qd[j] = 0.0f; }
        } else if (c->robot_reset_type == TF_RESET_RANDOM) {   /* trifinger_env.py:1125-1141 */
            float n[20];
            for (int b = 0; b < 5; ++b) rng4(c->seed, gid, count, RNG_ROBOT + (uint32_t)b, &n[4 * b]);
            for (int j = 0; j < 9; ++j) {
                e->q[j] = m->q_default[j % 3] + c->dof_pos_stddev * (2.0f * n[j] - 1.0f);
                e->qd[j] = 0.0f + c->dof_vel_stddev * (2.0f * n[9 + j] - 1.0f);
            }
        }
        if (c->object_reset_type == TF_RESET_DEFAULT) {
            e->cp[0] = h->ext ? 0.0f + e->dr[TF_DR_STAGE_POS] : 0.0f; e->cp[1] = h->ext ? 0.0f + e->dr[TF_DR_STAGE_POS + 1] : 0.0f; e->cp[2] = h->cfg.model.obj_min_height * e->dr[1];
            e->cq[0] = 0.0f; e->cq[1] = 0.0f; e->cq[2] = 0.0f; e->cq[3] = 1.0f;
            for (int k = 0; k < 3; ++k) { e->cv[k] = 0.0f; e->cw[k] = 0.0f; }
        } else if (c->object_reset_type == TF_RESET_RANDOM) {  /* trifinger_env.py:1169-1173 */
            float u[4];
            rng4(c->seed, gid, count, RNG_OBJECT, u);
            sample_xy(u[0], u[1], h->cfg.model.obj_max_com_dist, &e->cp[0], &e->cp[1]);
            if (h->ext) { e->cp[0] = e->cp[0] + e->dr[TF_DR_STAGE_POS]; e->cp[1] = e->cp[1] + e->dr[TF_DR_STAGE_POS + 1]; }   /* spawn relative to the stage */
            e->cp[2] = h->cfg.model.obj_min_height * e->dr[1];
            sample_yaw_quat(u[2], e->cq);
            for (int k = 0; k < 3; ++k) { e->cv[k] = 0.0f; e->cw[k] = 0.0f; }
        }
        sample_goal(h, gid, count, e);
        h->buf.reset_count[i] = count + 1u;
    }
    if (!force_all && h->buf.goal_reset_buf[i]) {
        uint32_t count = h->buf.reset_count[i];
        h->buf.goal_reset_buf[i] = 0;
        sample_goal(h, gid, count, e);
        h->buf.reset_count[i] = count + 1u;
    }
    return did_reset;
}

static void compute_torque(const struct TfHandle_* h, const float* act, const float q[9], const float qd[9],
                           float motor_scale, float tau[9]);
/* build-defined action repeat: keep the torque of the previous step with probability dr_action_repeat */
static void torque_with_repeat(const struct TfHandle_* h, int i, int64_t frame0, const float* act, Env* e) {
    const TfConfig* c = &h->cfg;
    float prev[9];
    for (int j = 0; j < 9; ++j) prev[j] = e->tau[j];
    compute_torque(h, act, e->q, e->qd, e->dr[3], e->tau);
    if (c->dr_enable && c->dr_action_repeat > 0.0f) {
        float u[4];
        rng4(c->seed, (uint32_t)(c->env_id_offset + i), (uint32_t)frame0, RNG_ACT_REPEAT, u);
        if (u[0] < c->dr_action_repeat) for (int j = 0; j < 9; ++j) e->tau[j] = prev[j];
    }
}

/* trifinger_env.py:442-494 */
static void compute_torque(const struct TfHandle_* h, const float* act, const float q[9], const float qd[9],
                           float motor_scale, float tau[9]) {
    const TfConfig* c = &h->cfg;
    int A = h->action_dim;
    float at[18];
    for (int j = 0; j < A; ++j) {
        if (c->normalize_action) {     /* torch_utils.py:39-57 */
            float off = (h->act_lo[j] + h->act_hi[j]) * 0.5f;
            at[j] = act[j] * (h->act_hi[j] - h->act_lo[j]) * 0.5f + off;
        } else at[j] = act[j];
    }
    for (int j = 0; j < 9; ++j) {
        float t;
        if (c->command_mode == TF_CMD_TORQUE) t = at[j];
        else if (c->command_mode == TF_CMD_POSITION) { t = h->kp[j] * (at[j] - q[j]); t = t - h->kd[j] * qd[j]; }
        else { t = at[9 + j] * (at[j] - q[j]); t = t - h->kd[j] * qd[j]; }
        t = f_max(f_min(t, 0.36f), -0.36f);
        if (c->apply_safety_damping) {
            t = t - h->ks[j] * qd[j];
            t = f_max(f_min(t, 0.36f), -0.36f);
        }
        tau[j] = t * motor_scale;     /* domain randomisation of the motor strength (1.0 when off) */
    }
}

typedef struct {
    float c_reach, c_move_pen, dt, c_dist, rot_num, rot_scale, w_rot, rot_delta_sched, w_rot_delta, w_move;
} RewardCoef;

static double sched_window(const TfRewardTerm* t, double step) {
    if (t->sched_start != t->sched_end) return (t->sched_start <= step && step <= t->sched_end) ? 1.0 : 0.0;
    return 1.0;
}
/* scalar prefactors exactly as python evaluates them in double before they meet an fp32 tensor */
static void reward_coefs(const struct TfHandle_* h, RewardCoef* rc) {
    const TfConfig* c = &h->cfg;
    double step = (double)h->frame_count * (double)c->global_num_envs;   /* env_base.py:287-289 */
    double dt = (double)c->dt;
    const TfRewardTerm* T = c->reward;
    rc->c_reach = (float)((double)T[TF_REW_FINGER_REACH_OBJECT_RATE].weight * sched_window(&T[TF_REW_FINGER_REACH_OBJECT_RATE], step));
    rc->c_move_pen = T[TF_REW_FINGER_MOVE_PENALTY].weight;
    rc->dt = c->dt;
    rc->c_dist = (float)((double)T[TF_REW_OBJECT_DIST].weight * dt * sched_window(&T[TF_REW_OBJECT_DIST], step));
    rc->rot_num = (float)(sched_window(&T[TF_REW_OBJECT_ROT], step) * dt);
    rc->rot_scale = c->object_rot_scale;
    rc->w_rot = T[TF_REW_OBJECT_ROT].weight;
    {
        const TfRewardTerm* t = &T[TF_REW_OBJECT_ROT_DELTA];   /* rewards.py:14-17 */
        double s = 1.0;
        if (t->sched_start != t->sched_end) {
            s = (step - t->sched_start) / (t->sched_end - t->sched_start);
            s = (s < 0.0) ? 0.0 : ((s > 1.0) ? 1.0 : s);
        }
        rc->rot_delta_sched = (float)s;
        rc->w_rot_delta = t->weight;
    }
    rc->w_move = T[TF_REW_OBJECT_MOVE].weight;
}

static float norm3d(const float a[3], const float b[3]) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return sqrtf(dx * dx + dy * dy + dz * dz);
}
/* torch.norm(a - b, p), reference rewards.py:216-226 (FingerReachObjectRatePenalty takes any p); p = 1, 2 and the maximum
 * norm are exact, an integer p in 3..16 takes the p-th root as exp(log(s) / p) plus one Newton step on y^p = s */
static float ipow(float x, int n) {
    float t = x;
    for (int k = 1; k < n; ++k) t = t * x;
    return t;
}
static float norm_p3(const float a[3], const float b[3], int p) {
    if (p == 2) return norm3d(a, b);
    const float ax = f_abs(a[0] - b[0]), ay = f_abs(a[1] - b[1]), az = f_abs(a[2] - b[2]);
    if (p == 1) return (ax + ay) + az;
    if (p == TF_NORM_INF) return f_max(f_max(ax, ay), az);
    /* the largest component is taken out first: d^p of a 4 mm distance underflows fp32 from p = 10 on */
    const float mx = f_max(f_max(ax, ay), az);
    if (!(mx > 0.0f)) return 0.0f;
    const float s = (ipow(ax / mx, p) + ipow(ay / mx, p)) + ipow(az / mx, p);      /* in [1, 3] */
    float y = tf_exp(tf_log(s) / (float)p);
    const float yp1 = ipow(y, p - 1);
    y = y - (yp1 * y - s) / ((float)p * yp1);
    return mx * y;
}

/* fingertip link state in the world frame: position, quaternion (xyzw), linear and angular velocity */
static void tip_state(const TfModel* m, int f, const float q[3], const float qd[3], float out[13]) {
    FK k;
    fk_setup(m, q, &k);
    float t[3], To[3];
    rot_link(&k, 3, m->tip_origin, t);
    To[0] = k.p3[0] + t[0]; To[1] = k.p3[1] + t[1]; To[2] = k.p3[2] + t[2];
    base_to_world(m, f, To, &out[0]);
    /* orientation: Rz(yaw) Ry(q1) Rx(q2+q3) */
    float sy, cy, sx, cx;
    tf_sincos(0.5f * q[0], &sy, &cy);
    tf_sincos(0.5f * (q[1] + q[2]), &sx, &cx);
    float qyx[4] = {cy * sx, sy * cx, -(sy * sx), cy * cx};   /* qy * qx */
    float qz[4] = {0.0f, 0.0f, m->base_half_yaw_sin[f], m->base_half_yaw_cos[f]};
    quat_mul(qz, qyx, &out[3]);
    float L1[3], L2[3], L3[3], vb[3], wb[3];
    levers(&k, To, L1, L2, L3);
    for (int i = 0; i < 3; ++i) vb[i] = L1[i] * qd[0] + L2[i] * qd[1] + L3[i] * qd[2];
    wb[0] = k.ax[0] * qd[1] + k.ax[0] * qd[2];
    wb[1] = qd[0];
    wb[2] = k.ax[2] * qd[1] + k.ax[2] * qd[2];
    dir_base_to_world(m, f, vb, &out[7]);
    dir_base_to_world(m, f, wb, &out[10]);
}

/* per-env accumulators for the info scalars */
typedef struct { double rew[6]; double pos_cnt, ori_cnt, succ, resets, nonfinite; } Stats;

/* trifinger_env.py:500-559 + 959-1099 for one env.  prev_obj = history[1] pose (7).  */
/* scale_transform (torch_utils.py:18-36: 2 (x - offset) / (upper - lower)) as ONE fused multiply-add with the two
 * constants of the slot, k1 = 2 / range and k0 = -(2 offset) / range, each rounded to fp32 once: x k1 + k0.  Differs
 * from the reference's three roundings by at most a few ulp of the scaled value (< 1e-6; the golden fixtures pin it). */
static inline float scale_slot(float x, float off, float inv) {
    const float k1 = 2.0f * inv, k0 = -(2.0f * off) * inv;
    return FMA(x, k1, k0);
}
static void post_step_env(const struct TfHandle_* h, int i, Env* e, const float prev_obj[7], const RewardCoef* rc,
                          int with_reward, Stats* st) {
    const TfConfig* c = &h->cfg;
    const TfModel* m = &c->model;
    int A = h->action_dim, OD = h->obs_dim, SD = h->states_dim;
    float tips[3][13];
    for (int f = 0; f < 3; ++f) {
        tip_state(m, f, &e->q[3 * f], &e->qd[3 * f], tips[f]);
        if (h->ext) for (int j = 0; j < 3; ++j) tips[f][j] = tips[f][j] + e->dr[TF_DR_BASE_POS + j];     /* robot frame -> world */
    }
    /* NaN guard: a non-finite env is flagged for reset and parked at the default pose; its reward terms of this step
     * (they involve the histories that were non-finite) are zero, so neither the learner nor the logged means see it */
    int guarded = 0;
    {
        float acc = 0.0f;
        for (int j = 0; j < 9; ++j) acc = acc + e->q[j] * 0.0f + e->qd[j] * 0.0f;
        for (int j = 0; j < 3; ++j) acc = acc + e->cp[j] * 0.0f + e->cv[j] * 0.0f + e->cw[j] * 0.0f;
        for (int j = 0; j < 4; ++j) acc = acc + e->cq[j] * 0.0f;
        if (!(acc == 0.0f)) {
            for (int j = 0; j < 9; ++j) { e->q[j] = m->q_default[j % 3]; e->qd[j] = 0.0f; }
            e->cp[0] = 0.0f; e->cp[1] = 0.0f; e->cp[2] = h->cfg.model.obj_min_height;
            e->cq[0] = 0.0f; e->cq[1] = 0.0f; e->cq[2] = 0.0f; e->cq[3] = 1.0f;
            for (int k = 0; k < 3; ++k) { e->cv[k] = 0.0f; e->cw[k] = 0.0f; }
            for (int j = 0; j < 18; ++j) e->ft[j] = 0.0f;
            for (int f = 0; f < 3; ++f) {
                tip_state(m, f, &e->q[3 * f], &e->qd[3 * f], tips[f]);
                if (h->ext) for (int j = 0; j < 3; ++j) tips[f][j] = tips[f][j] + e->dr[TF_DR_BASE_POS + j];
            }
            h->buf.reset_buf[i] = 1;
            st->nonfinite += 1.0;
            guarded = 1;
        }
    }
    /* observations (trifinger_env.py:996-1019) and states (:1021-1051) */
    float raw[MAX_STATES];
    int k = 0;
    for (int j = 0; j < 9; ++j) raw[k++] = e->q[j];
    for (int j = 0; j < 9; ++j) raw[k++] = e->qd[j];
    for (int j = 0; j < 3; ++j) raw[k++] = e->cp[j];
    for (int j = 0; j < 4; ++j) raw[k++] = e->cq[j];
    for (int j = 0; j < 3; ++j) raw[k++] = e->gp[j];
    for (int j = 0; j < 4; ++j) raw[k++] = e->gq[j];
    const float* act = &h->buf.action_buf[(size_t)i * (size_t)A];
    for (int j = 0; j < A; ++j) raw[k++] = act[j];
    float* obs = &h->buf.obs[(size_t)i * (size_t)OD];
    /* every emitted value passes the fused wrapper clipping (a no-op when off) */
    for (int j = 0; j < OD; ++j) obs[j] = f_clamp(c->normalize_obs ? scale_slot(raw[j], h->obs_off[j], h->obs_inv[j]) : raw[j], -h->clip_obs, h->clip_obs);
    if (c->dr_enable && c->dr_obs_noise > 0.0f) {      /* observation noise on q, qd, object pose (slots 0..24) */
        uint32_t gid = (uint32_t)(c->env_id_offset + i);
        float nz[28];
        for (int b = 0; b < 7; ++b) rng4(c->seed, gid, (uint32_t)h->frame_count, RNG_OBS_NOISE + (uint32_t)b, &nz[4 * b]);
        for (int j = 0; j < 25; ++j) obs[j] = f_clamp(FMA(c->dr_obs_noise, 2.0f * nz[j] - 1.0f, obs[j]), -h->clip_obs, h->clip_obs);
    }
    if (c->asymmetric_obs) {
        for (int j = 0; j < 3; ++j) raw[k++] = e->cv[j];
        for (int j = 0; j < 3; ++j) raw[k++] = e->cw[j];
        for (int f = 0; f < 3; ++f) for (int j = 0; j < 13; ++j) raw[k++] = tips[f][j];
        for (int j = 0; j < 9; ++j) raw[k++] = c->enable_ft_sensors ? e->tau[j] : 0.0f;
        {
            /* wrench: mean over the substeps of this step, rotated into the tip-link frame */
            float inv_n = 1.0f / (float)(c->substeps * c->control_decimation);
            for (int f = 0; f < 3; ++f) {
                FK kk;
                fk_setup(m, &e->q[3 * f], &kk);
                for (int half = 0; half < 2; ++half) {
                    float wv[3] = {e->ft[6 * f + 3 * half] * inv_n, e->ft[6 * f + 3 * half + 1] * inv_n,
                                   e->ft[6 * f + 3 * half + 2] * inv_n};
                    float bv[3], lv[3];
                    dir_world_to_base(m, f, wv, bv);
                    rot_link_T(&kk, 3, bv, lv);
                    for (int j = 0; j < 3; ++j) raw[k++] = c->enable_ft_sensors ? lv[j] : 0.0f;
                }
            }
        }
        float* sts = &h->buf.states[(size_t)i * (size_t)SD];
        for (int j = 0; j < SD; ++j) sts[j] = f_clamp(c->normalize_obs ? scale_slot(raw[j], h->st_off[j], h->st_inv[j]) : raw[j], -h->clip_obs, h->clip_obs);
    }
    /* history bookkeeping: previous fingertip positions are whatever the last filled frame left */
    float tip_prev[9];
    for (int j = 0; j < 9; ++j) tip_prev[j] = ST(h, TF_S_TIP_P + j, i);
    for (int f = 0; f < 3; ++f) for (int j = 0; j < 3; ++j) ST(h, TF_S_TIP_P + 3 * f + j, i) = tips[f][j];
    if (!with_reward) return;
    /* rewards (rewards.py; evaluation order trifinger_env.py:513-550) */
    float r[6];
    {
        float s = 0.0f;
        for (int f = 0; f < 3; ++f) {
            float cur = norm_p3(tips[f], e->cp, c->finger_reach_norm_p);
            float prv = norm_p3(&tip_prev[3 * f], prev_obj, c->finger_reach_norm_p);
            s = s + (cur - prv);
        }
        r[0] = rc->c_reach * s;
    }
    {
        float s = 0.0f;
        for (int f = 0; f < 3; ++f) for (int j = 0; j < 3; ++j) {
            float vel = (tips[f][j] - tip_prev[3 * f + j]) / rc->dt;
            s = s + vel * vel;
        }
        r[1] = rc->c_move_pen * s;
    }
    float dist = norm3d(e->cp, e->gp);
    r[2] = rc->c_dist * lgsk(dist, 50.0f);
    float ang = quat_diff_rad(e->cq, e->gq);
    r[3] = rc->w_rot * (rc->rot_num / (rc->rot_scale * f_abs(ang) + rc->rot_scale));
    float ang_prev = quat_diff_rad(&prev_obj[3], e->gq);
    r[4] = rc->w_rot_delta * (rc->rot_delta_sched * (f_abs(ang) - f_abs(ang_prev)));
    r[5] = rc->w_move * (dist - norm3d(prev_obj, e->gp));
    if (guarded) for (int t = 0; t < 6; ++t) r[t] = 0.0f;
    float total = 0.0f;
    for (int t = 0; t < 6; ++t) if (c->reward[t].activate) { total = total + r[t]; st->rew[t] += (double)r[t]; }
    /* termination (trifinger_env.py:1053-1099) */
    int pos_ok = dist <= c->position_tolerance;
    int ori_ok = ang <= c->orientation_tolerance;
    st->pos_cnt += pos_ok; st->ori_cnt += ori_ok;
    int done;
    if (c->task_difficulty < 4) done = pos_ok;
    else if (c->task_difficulty == 4) done = pos_ok && ori_ok;
    else done = ori_ok;
    int succ = h->buf.successes[i] != 0;
    if (c->success_activate) {
        if (done) total = total + c->success_bonus;
        h->buf.goal_reset_buf[i] = (uint8_t)done;
        succ = succ || done;
    } else {
        succ = (h->buf.goal_reset_buf[i] != 0) && succ;
    }
    h->buf.successes[i] = (uint8_t)succ;
    st->succ += succ;
    h->buf.reward[i] = total;
}

/* env_base.py:391-399 */
static void finish_env(const struct TfHandle_* h, int i) {
    int s = (int)h->buf.steps[i] + 1;
    h->buf.steps[i] = s;
    if (h->cfg.episode_length > 0 && s >= h->cfg.episode_length) h->buf.reset_buf[i] = 1;
    h->buf.dones[i] = (uint8_t)(h->buf.reset_buf[i] && h->buf.goal_reset_buf[i]);
}

static void write_info(const struct TfHandle_* h, const Stats* st) {
    float n = (float)h->cfg.num_envs;
    for (int t = 0; t < 6; ++t) h->buf.info[t] = (float)(st->rew[t] / (double)n);
    h->buf.info[TF_INFO_POS_COUNT] = (float)st->pos_cnt;
    h->buf.info[TF_INFO_ORI_COUNT] = (float)st->ori_cnt;
    h->buf.info[TF_INFO_SUCCESS_MEAN] = (float)(st->succ / (double)n);
    h->buf.info[TF_INFO_NUM_RESETS] = (float)st->resets;
    h->buf.info[TF_INFO_NUM_NONFINITE] = (float)st->nonfinite;
}

static void stats_add(Stats* a, const Stats* b) {
    for (int t = 0; t < 6; ++t) a->rew[t] += b->rew[t];
    a->pos_cnt += b->pos_cnt; a->ori_cnt += b->ori_cnt; a->succ += b->succ; a->resets += b->resets;
    a->nonfinite += b->nonfinite;
}

/* fused step / reset */
/* action == NULL && random_actions: every env draws its action 2 u - 1 (tf_step_random) */
static int run_step(tf_handle h, const float* action, int is_reset, int random_actions) {
    if (!h) return TF_ERR_INVALID_ARG;
    if (!h->bound) return TF_ERR_NOT_BOUND;
    if (!is_reset && !action && !random_actions) return TF_ERR_INVALID_ARG;
    const TfConfig* c = &h->cfg;
    int N = c->num_envs, A = h->action_dim;
    int nsim = is_reset ? 1 : c->control_decimation;
    h->frame_count += nsim;
    RewardCoef rc;
    reward_coefs(h, &rc);
    float hsub = c->dt / (float)c->substeps;
    Stats total;
    memset(&total, 0, sizeof(total));
#pragma omp parallel
    {
        Stats local;
        memset(&local, 0, sizeof(local));
#pragma omp for schedule(static)
        for (int i = 0; i < N; ++i) {
            Env e;
            env_load(h, i, &e);
            float* abuf = &h->buf.action_buf[(size_t)i * (size_t)A];
            if (is_reset) { for (int j = 0; j < A; ++j) abuf[j] = 0.0f; }   /* env_base.py:332-334 acts on the buffer */
            else if (random_actions) {
                for (int b = 0; b < (A + 3) / 4; ++b) {
                    float u[4];
                    rng4(c->seed, (uint32_t)(c->env_id_offset + i), (uint32_t)(h->frame_count - nsim), RNG_ACTION + (uint32_t)b, u);
                    for (int k = 0; k < 4 && 4 * b + k < A; ++k) abuf[4 * b + k] = f_clamp(2.0f * u[k] - 1.0f, -h->clip_act, h->clip_act);
                }
            }
            else { for (int j = 0; j < A; ++j) abuf[j] = f_clamp(action[(size_t)i * (size_t)A + j], -h->clip_act, h->clip_act); }
            if (apply_resets(h, i, &e, is_reset)) {
                for (int j = 0; j < A; ++j) abuf[j] = 0.0f;                  /* trifinger_env.py:387 */
                local.resets += 1.0;
            }
            if (is_reset) compute_torque(h, abuf, e.q, e.qd, e.dr[3], e.tau);
            else torque_with_repeat(h, i, h->frame_count - nsim, abuf, &e);
            float prev_obj[7] = {e.cp[0], e.cp[1], e.cp[2], e.cq[0], e.cq[1], e.cq[2], e.cq[3]};
            for (int j = 0; j < 3; ++j) ST(h, TF_S_PREV_OBJ_P + j, i) = e.cp[j];   /* history[1] of the object */
            for (int j = 0; j < 4; ++j) ST(h, TF_S_PREV_OBJ_Q + j, i) = e.cq[j];
            for (int j = 0; j < 18; ++j) e.ft[j] = 0.0f;
            for (int s = 0; s < nsim * c->substeps; ++s) substep(h, &e, hsub);
            post_step_env(h, i, &e, prev_obj, &rc, !is_reset, &local);
            goal_advance(h, &e, nsim * c->substeps, hsub);
            env_store(h, i, &e, 0);          /* the fused step keeps the fingertip wrench of the step to itself: TF_S_FT is split-path state */
            if (!is_reset) {
                finish_env(h, i);
                if (h->buf.reset_buf[i]) presample_next_reset(h, i);
            }
        }
#pragma omp critical
        stats_add(&total, &local);
    }
    write_info(h, &total);
    return TF_OK;
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
int tf_enable_kernel_timing(tf_handle h, int32_t max_launches) {
    if (!h) return TF_ERR_INVALID_ARG;
    h->timing_on = max_launches > 0; h->timed_ms = 0.0; h->timed_launches = 0;
    return TF_OK;
}
int tf_set_kernel_timing_window(tf_handle h, int32_t window) {
    if (!h || window <= 0) return TF_ERR_INVALID_ARG;
    return TF_OK;                       /* host timing costs nothing: every launch stays timed */
}
int tf_kernel_time_ms(tf_handle h, double* total_ms, int64_t* launches) {
    if (!h || !total_ms || !launches) return TF_ERR_INVALID_ARG;
    *total_ms = h->timed_ms; *launches = h->timed_launches;
    return TF_OK;
}
int tf_step(tf_handle h, const float* action, void* stream) {
    (void)stream;
    double t0 = (h && h->timing_on) ? now_ms() : 0.0;
    int rc = run_step(h, action, 0, 0);
    if (h && h->timing_on) { h->timed_ms += now_ms() - t0; h->timed_launches += 1; }
    return rc;
}
int tf_step_random(tf_handle h, void* stream) {
    (void)stream;
    double t0 = (h && h->timing_on) ? now_ms() : 0.0;
    int rc = run_step(h, NULL, 0, 1);
    if (h && h->timing_on) { h->timed_ms += now_ms() - t0; h->timed_launches += 1; }
    return rc;
}
int tf_reset(tf_handle h, void* stream) { (void)stream; return run_step(h, NULL, 1, 0); }

/* ---- split path ---- */
int tf_apply_resets(tf_handle h, void* stream) {
    (void)stream;
    if (!h) return TF_ERR_INVALID_ARG;
    if (!h->bound) return TF_ERR_NOT_BOUND;
    int A = h->action_dim;
    for (int i = 0; i < h->cfg.num_envs; ++i) {
        Env e;
        env_load(h, i, &e);
        float* ab = &h->buf.action_buf[(size_t)i * (size_t)A];
        for (int j = 0; j < A; ++j) ab[j] = f_clamp(ab[j], -h->clip_act, h->clip_act);
        if (apply_resets(h, i, &e, 0)) for (int j = 0; j < A; ++j) ab[j] = 0.0f;
        env_store(h, i, &e, 1);
    }
    return TF_OK;
}
int tf_pre_step(tf_handle h, void* stream) {
    (void)stream;
    if (!h) return TF_ERR_INVALID_ARG;
    if (!h->bound) return TF_ERR_NOT_BOUND;
    int A = h->action_dim;
    for (int i = 0; i < h->cfg.num_envs; ++i) {
        Env e;
        env_load(h, i, &e);
        torque_with_repeat(h, i, h->frame_count, &h->buf.action_buf[(size_t)i * (size_t)A], &e);
        for (int j = 0; j < 18; ++j) e.ft[j] = 0.0f;
        env_store(h, i, &e, 1);
        for (int j = 0; j < 3; ++j) ST(h, TF_S_PREV_OBJ_P + j, i) = e.cp[j];
        for (int j = 0; j < 4; ++j) ST(h, TF_S_PREV_OBJ_Q + j, i) = e.cq[j];
    }
    return TF_OK;
}
int tf_simulate(tf_handle h, void* stream) {
    (void)stream;
    if (!h) return TF_ERR_INVALID_ARG;
    if (!h->bound) return TF_ERR_NOT_BOUND;
    h->frame_count += 1;
    float hsub = h->cfg.dt / (float)h->cfg.substeps;
    for (int i = 0; i < h->cfg.num_envs; ++i) {
        Env e;
        env_load(h, i, &e);
        for (int s = 0; s < h->cfg.substeps; ++s) substep(h, &e, hsub);
        env_store(h, i, &e, 1);
    }
    return TF_OK;
}
int tf_post_step(tf_handle h, void* stream) {
    (void)stream;
    if (!h) return TF_ERR_INVALID_ARG;
    if (!h->bound) return TF_ERR_NOT_BOUND;
    RewardCoef rc;
    reward_coefs(h, &rc);
    Stats st;
    memset(&st, 0, sizeof(st));
    for (int i = 0; i < h->cfg.num_envs; ++i) {
        Env e;
        env_load(h, i, &e);
        float prev_obj[7];
        for (int j = 0; j < 3; ++j) prev_obj[j] = ST(h, TF_S_PREV_OBJ_P + j, i);
        for (int j = 0; j < 4; ++j) prev_obj[3 + j] = ST(h, TF_S_PREV_OBJ_Q + j, i);
        post_step_env(h, i, &e, prev_obj, &rc, 1, &st);
        goal_advance(h, &e, h->cfg.control_decimation * h->cfg.substeps, h->cfg.dt / (float)h->cfg.substeps);
        env_store(h, i, &e, 1);
    }
    write_info(h, &st);
    return TF_OK;
}
int tf_finish_step(tf_handle h, void* stream) {
    (void)stream;
    if (!h) return TF_ERR_INVALID_ARG;
    if (!h->bound) return TF_ERR_NOT_BOUND;
    for (int i = 0; i < h->cfg.num_envs; ++i) finish_env(h, i);
    return TF_OK;
}

/* ---- leaf entries for the golden tests ---- */
int tf_test_quat_diff_rad(const float* a, const float* b, float* out, int32_t n, void* s) {
    (void)s;
    for (int i = 0; i < n; ++i) out[i] = quat_diff_rad(&a[4 * i], &b[4 * i]);
    return TF_OK;
}
int tf_test_quat_mul(const float* a, const float* b, float* out, int32_t n, void* s) {
    (void)s;
    for (int i = 0; i < n; ++i) quat_mul(&a[4 * i], &b[4 * i], &out[4 * i]);
    return TF_OK;
}
int tf_test_lgsk(const float* x, float scale, float* out, int32_t n, void* s) {
    (void)s;
    for (int i = 0; i < n; ++i) out[i] = lgsk(x[i], scale);
    return TF_OK;
}
int tf_test_sample_xy(const float* ur, const float* ut, float r_max, float* x, float* y, int32_t n, void* s) {
    (void)s;
    for (int i = 0; i < n; ++i) sample_xy(ur[i], ut[i], r_max, &x[i], &y[i]);
    return TF_OK;
}
int tf_test_sample_yaw_quat(const float* u, float* quat, int32_t n, void* s) {
    (void)s;
    for (int i = 0; i < n; ++i) sample_yaw_quat(u[i], &quat[4 * i]);
    return TF_OK;
}
int tf_test_normalize_quat(const float* nn, float* quat, int32_t n, void* s) {
    (void)s;
    for (int i = 0; i < n; ++i) normalize_quat(&nn[4 * i], &quat[4 * i]);
    return TF_OK;
}
int tf_test_philox(uint64_t seed, const uint32_t* env_id, const uint32_t* counter, uint32_t tag, uint32_t* out4,
                   int32_t n, void* s) {
    (void)s;
    for (int i = 0; i < n; ++i)
        philox4x32_10(env_id[i], counter[i], tag, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), &out4[4 * i]);
    return TF_OK;
}
int tf_test_finger_dynamics(tf_handle h, const float* q, const float* qd, float* tip, float* mass, float* bias,
                            int32_t n, void* s) {
    (void)s;
    if (!h) return TF_ERR_INVALID_ARG;
    const TfModel* m = &h->cfg.model;
    for (int i = 0; i < n; ++i) {
        FK k;
        float M[6], t[3];
        fk_setup(m, &q[3 * i], &k);
        finger_dynamics(m, &k, &qd[3 * i], h->cfg.gravity, M, &bias[3 * i]);
        rot_link(&k, 3, m->tip_origin, t);
        for (int j = 0; j < 3; ++j) tip[3 * i + j] = k.p3[j] + t[j];
        float* o = &mass[9 * i];
        o[0] = M[0]; o[1] = M[1]; o[2] = M[2]; o[3] = M[1]; o[4] = M[3]; o[5] = M[4]; o[6] = M[2]; o[7] = M[4]; o[8] = M[5];
    }
    return TF_OK;
}

/* thread control for the OpenMP build (bench.py's cpu_baseline): set the team size, return what will be used */
/* test entry (tests/test_model_fixture.py): gap and link of the finger-cube contact candidate of finger f for joint angles q and a cube pose (world),
 * exactly as the substep selects it (default model geometry of `m`, no domain randomisation) */
void tfo_finger_gap(const TfModel* m, int32_t f, const float q[3], const float cube_p[3], const float cube_q[4], float* gap_out, int32_t* link_out) {
    FK k;
    float Ab[3], Bb[3], Aw[3], Bw[3], R[9], hc[3], x[3], y[3], nc[3], gap, radius;
    int link;
    fk_setup(m, q, &k);
    link_point(&k, 3, m->cap_a, Ab);
    link_point(&k, 3, m->cap_b, Bb);
    base_to_world(m, f, Ab, Aw);
    base_to_world(m, f, Bb, Bw);
    quat_to_rot(cube_q, R);
    for (int i = 0; i < 3; ++i) hc[i] = m->box ? m->box_half[i] : m->cube_half;
    finger_cube_candidates(m, f, &k, Aw, Bw, cube_p, R, hc, cube_p[2], &gap, x, y, nc, &radius, &link);
    *gap_out = gap;
    *link_out = link;
}

int tfo_omp_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* extra oracle-only leaf functions used by tests/test_oracle_math.py */
void tfo_sincos(const float* x, float* s, float* c, int32_t n) { for (int i = 0; i < n; ++i) tf_sincos(x[i], &s[i], &c[i]); }
void tfo_exp(const float* x, float* y, int32_t n) { for (int i = 0; i < n; ++i) y[i] = tf_exp(x[i]); }
void tfo_asin(const float* x, float* y, int32_t n) { for (int i = 0; i < n; ++i) y[i] = tf_asin(x[i]); }
void tfo_log(const float* x, float* y, int32_t n) { for (int i = 0; i < n; ++i) y[i] = tf_log(x[i]); }
void tfo_rcp(const float* x, float* y, int32_t n) { for (int i = 0; i < n; ++i) y[i] = f_rcp(x[i]); }
void tfo_rsqrt(const float* x, float* y, int32_t n) { for (int i = 0; i < n; ++i) y[i] = f_rsqrt(x[i]); }
void tfo_philox_raw(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
