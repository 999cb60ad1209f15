// trifinger_hip.hip - MI355X (gfx950) kernels + C ABI of the TriFinger vectorised environment step.
//
// The hot path of pairlab/leibnizgym (IsaacEnvBase.step -> TrifingerEnv hooks -> gymapi.simulate;
// reference leibnizgym/envs/env_base.py:345-401, leibnizgym/envs/trifinger/trifinger_env.py:373-559,959-1265)
// as ONE fused launch per control step:
//
//   masked reset / goal reset (Philox4x32-10 keyed by global env id)  ->  PD/torque law  ->
//   decimation x substeps x { 3 x 3-DoF articulated forward dynamics + free cube, contact generation (link capsules,
//   finger-finger, fingertip and cube against floor and stepped boundary), warm-started projected Gauss-Seidel,
//   symplectic Euler }  ->  fingertip FK, obs[41]/states[113] assembly + normalisation, six reward terms, termination,
//   step counters / time-out / dones, wave-reduced episode statistics.
//
// Execution model (CDNA4): a workgroup of 4 wavefronts owns 64 environments, one per lane; wavefronts 0..2 are the three
// fingers, wavefront 3 is the cube (tf_roles.h).  The four roles run concurrently on the four SIMDs of a CU and exchange
// a few floats per lane through LDS at workgroup barriers; 4 workgroups per CU put 4 wavefronts on every SIMD (<= 128
// registers each), which hides the dependent-issue latency a single 512-register wavefront per SIMD (the round-1
// design) was bound by.  State lives in HBM as structure-of-arrays rows [field][env]: every global access of a wavefront
// is one coalesced 256-B line.  Per-env matrices are at most 3x3 / 6x6: no MFMA.  The row-major API tensors
// (action [N,A], obs [N,41], states [N,113]) are transposed through LDS by the whole workgroup (dwordx4, coalesced).
// Episode statistics are reduced with DPP butterflies per wavefront and folded across workgroups with fixed-point integer
// atomics (order independent, hence deterministic; the last arriving workgroup writes info[]): one launch, no host sync.
//
// Physics is this build's own spec (the reference's lives in closed-source PhysX): DESIGN.md "Physics spec".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>

#include "tf_roles.h"
#include "../../include/trifinger_default_caps.h"

#define SCR_STRIDE TF_SCR_STRIDE

#include "tf_launch.h"

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
    int variant;             // TF_KERNEL_AUTO / NARROW / WIDE / WIDE_HELPERS as asked for (tf_set_kernel_variant)
    bool wide;               // what the launches use
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
    {   // shapes of the three links: include/trifinger_default_caps.h (fitted to the collision hulls by tools/fit_link_shapes.py)
        static const TfLinkShape sh3 = TF_DEFAULT_SHAPE3, sh2 = TF_DEFAULT_SHAPE2, sh1 = TF_DEFAULT_SHAPE1;
        static const TfSphere s3[1] = { TF_DEFAULT_SPH3 }, s2[2] = { TF_DEFAULT_SPH2 };
        m->shape3 = sh3; m->shape2 = sh2; m->shape1 = sh1;
        m->sph3[0] = s3[0]; m->sph2[0] = s2[0]; m->sph2[1] = s2[1];
    }
    m->upper_check_z = 0.17f;
    m->middle_check_z = 0.075f;
    m->cube_half = 0.0325f;
    m->cube_mass = (float)(291.3 * 0.065 * 0.065 * 0.065);
    m->cube_inertia = (float)(291.3 * 0.065 * 0.065 * 0.065 * 0.065 * 0.065 / 6.0);
    m->cube_linear_damping = 0.0f;
    m->cube_angular_damping = 0.05f;
    // boundary profile (knots of a piecewise-linear r(z)): vertical ring up to 32 mm, then the flaring cone of the stage; radii = mid-way
    // between the chords and the corners of the 40 convex pieces (tests/golden/model.npz: boundary_profile_*; tests/test_model_fixture.py)
    m->wall_r[0] = 0.1895f; m->wall_z[0] = 0.032f;
    m->wall_r[1] = 0.2093f; m->wall_z[1] = 0.06f;
    m->wall_r[2] = 0.2319f; m->wall_z[2] = 0.10f;
    m->wall_r[3] = 0.2741f; m->wall_z[3] = 0.176f;
    m->mu_finger_cube = 1.0f;
    m->mu_cube_floor = 0.55f;
    m->mu_tip_floor = 0.55f;
    m->mu_cube_wall = 1.0f;
    m->mu_tip_wall = 1.0f;
    m->mu_finger_finger = 1.0f;
    m->mu_robot = 1.0f; m->mu_object = 1.0f; m->mu_floor = 0.1f; m->mu_stage = 1.0f;   // trifinger_env.py:364-365,876-878,914-915,934-936
    m->restitution_finger = 0.4f;
    m->restitution_ff = 0.8f;
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
    m->obj_radius_3d = 0.05629165f;       // CuboidalObject(0.065): reference envs/trifinger/utils.py:122-131
    m->obj_max_com_dist = 0.13870835f;
    m->obj_min_height = 0.0325f;
    m->obj_span_min_height = 0.0675f;
    m->obj_span_radius = 0.04370835f;
    m->ff_middle_pairs = 1;                // API 8: the reference keeps every robot link in one self-colliding group (trifinger_env.py:811-812)
}

// The object as a general box: mass, principal moments about the body axes, reference inertia of the scaled solve (the mean
// of the principal moments), CuboidalObject constants (reference envs/trifinger/utils.py:122-131, ARENA_RADIUS :54).
void tf_model_set_box(TfModel* m, const float size[3], float density) {
    const double sx = size[0], sy = size[1], sz = size[2];
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
    if (cfg->num_envs > TF_MAX_ENVS) return TF_ERR_INVALID_ARG;     // 32-bit buffer offsets into state[TF_STATE_ROWS][N]
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
    TfHandle_* h = new TfHandle_();
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->ev_stride = 1;
    if (h->cfg.global_num_envs <= 0) h->cfg.global_num_envs = cfg->num_envs;
    h->action_dim = tf_action_dim(cfg->command_mode);
    h->variant = TF_KERNEL_AUTO;
    h->wide = cfg->num_envs <= TF_WIDE_MAX_ENVS;
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
    P.norm_p = cfg->finger_reach_norm_p;
    P.robot_reset_type = cfg->robot_reset_type; P.object_reset_type = cfg->object_reset_type;
    P.goal_rotation_activate = cfg->goal_rotation_activate;
    P.dr_enable = cfg->dr_enable;
    for (int i = 0; i < 2; ++i) {
        P.dr_cube_mass[i] = cfg->dr_cube_mass[i]; P.dr_cube_size[i] = cfg->dr_cube_size[i]; P.dr_friction[i] = cfg->dr_friction[i];
        P.dr_motor[i] = cfg->dr_motor[i]; P.dr_link_mass[i] = cfg->dr_link_mass[i]; P.dr_restitution[i] = cfg->dr_restitution[i];
        P.dr_friction_robot[i] = cfg->dr_friction_robot[i]; P.dr_friction_object[i] = cfg->dr_friction_object[i];
        P.dr_friction_stage[i] = cfg->dr_friction_stage[i]; P.dr_stage_pos[i] = cfg->dr_stage_pos[i];
    }
    for (int i = 0; i < 3; ++i) P.dr_base_pos[i] = cfg->dr_base_pos[i];
    P.dr_obs_noise = (cfg->dr_enable && cfg->dr_obs_noise > 0.0f) ? cfg->dr_obs_noise : 0.0f;
    P.dr_action_repeat = (cfg->dr_enable && cfg->dr_action_repeat > 0.0f) ? cfg->dr_action_repeat : 0.0f;
    P.clip_obs = 3.402823466e38f; P.clip_act = 3.402823466e38f;
    P.dof_pos_stddev = cfg->dof_pos_stddev; P.dof_vel_stddev = cfg->dof_vel_stddev; P.goal_rate = cfg->goal_rotation_rate_magnitude;
    for (int t = 0; t < 6; ++t) P.rew_active[t] = cfg->reward[t].activate;
    P.success_activate = cfg->success_activate; P.success_bonus = cfg->success_bonus;
    P.pos_tol = cfg->position_tolerance; P.ori_tol = cfg->orientation_tolerance;
    P.substeps = cfg->substeps; P.iters = cfg->solver_iterations * cfg->solver_inner; P.inner = cfg->solver_inner; P.control_decimation = cfg->control_decimation;
    P.dt = cfg->dt; P.hsub = cfg->dt / (float)cfg->substeps;
    for (int i = 0; i < 3; ++i) P.grav[i] = cfg->gravity[i];
    P.m = cfg->model;
    for (int i = 0; i < 3; ++i) {
        const double sl = ((double)P.m.wall_r[i + 1] - (double)P.m.wall_r[i]) / ((double)P.m.wall_z[i + 1] - (double)P.m.wall_z[i]);
        P.wall_s[i] = (float)sl;
        P.wall_c[i] = (float)(1.0 / sqrt(1.0 + sl * sl));
        P.wall_sn[i] = (float)(sl / sqrt(1.0 + sl * sl));
    }
    void* tk = nullptr;
    e = hipMalloc(&tk, STAT_WORDS * sizeof(unsigned long long));
    if (e != hipSuccess) { delete h; return hip_fail(e, "hipMalloc(tickets)"); }
    P.tickets = (gu32*)tk;
    e = hipMemset(tk, 0, STAT_WORDS * sizeof(unsigned long long));
    if (e != hipSuccess) { (void)hipFree((void*)P.tickets); delete h; return hip_fail(e, "hipMemset(tickets)"); }
    e = hipMalloc((void**)&h->d_params, sizeof(DevParams));
    if (e != hipSuccess) { (void)hipFree((void*)P.tickets); delete h; return hip_fail(e, "hipMalloc(params)"); }
    e = hipMemcpy(h->d_params, &h->dp, sizeof(DevParams), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->d_params); (void)hipFree((void*)P.tickets); delete h; return hip_fail(e, "hipMemcpy(params)"); }
    *out = h;
    return TF_OK;
}

// Cold-path rewrite of the device parameter block.  Steps may be in flight on ANY stream (torch side streams are
// non-blocking: a copy on the null stream does not order behind them), so the device is drained first.
static hipError_t push_params(TfHandle_* h) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    return hipMemcpy(h->d_params, &h->dp, sizeof(DevParams), hipMemcpyHostToDevice);
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
    P.dones = (gu8*)b->dones; P.steps = (gi64*)b->steps; P.reset_count = (gu32*)b->reset_count;
    P.info = (gfloat*)b->info; P.scratch = (gfloat*)b->scratch;
    HIP_TRY(push_params(h));
    h->bound = 1;
    return TF_OK;
}

int tf_set_clipping(tf_handle h, float clip_obs, float clip_actions) {
    if (!h) return TF_ERR_INVALID_ARG;
    h->dp.clip_obs = (clip_obs > 0.0f) ? clip_obs : 3.402823466e38f;
    h->dp.clip_act = (clip_actions > 0.0f) ? clip_actions : 3.402823466e38f;
    HIP_TRY(push_params(h));
    return TF_OK;
}
int tf_set_gravity(tf_handle h, const float g[3]) {
    if (!h || !g) return TF_ERR_INVALID_ARG;
    for (int i = 0; i < 3; ++i) { h->cfg.gravity[i] = g[i]; h->dp.grav[i] = g[i]; }
    HIP_TRY(push_params(h));
    return TF_OK;
}
static int ext_kind(const TfConfig& c);
// helper wavefronts (the WIDE = 2 units): asked for, or picked for a population that leaves every CU to one workgroup (with the fast contact set too: the
// distal pass on wavefront 7 alone is worth 1.4 us at 8192 envs)
static bool use_helpers(const TfHandle_* h) {
    if (!h->wide) return false;
    if (h->variant == TF_KERNEL_WIDE_HELPERS) return true;
    return h->variant == TF_KERNEL_AUTO && h->cfg.num_envs <= TF_HELPERS_MAX_ENVS;
}
int tf_set_kernel_variant(tf_handle h, int32_t variant) {
    if (!h || variant < TF_KERNEL_AUTO || variant > TF_KERNEL_WIDE_HELPERS) return TF_ERR_INVALID_ARG;
    h->variant = variant;
    h->wide = (variant == TF_KERNEL_AUTO) ? (h->cfg.num_envs <= TF_WIDE_MAX_ENVS) : (variant != TF_KERNEL_NARROW);
    return TF_OK;
}
int tf_kernel_variant(tf_handle h) { return h ? (h->wide ? (use_helpers(h) ? TF_KERNEL_WIDE_HELPERS : TF_KERNEL_WIDE) : TF_KERNEL_NARROW) : TF_ERR_INVALID_ARG; }
int tf_kernel_occupancy(tf_handle h) {
    if (!h) return TF_ERR_INVALID_ARG;
    const bool asym = h->cfg.asymmetric_obs != 0;
    const bool help = use_helpers(h);
#if defined(TF_DEV_MIN)
    if (help) return tf_occupancy_env_0_2(h->action_dim, asym);
    return h->wide ? tf_occupancy_env_0_1(h->action_dim, asym) : tf_occupancy_env_0_0(h->action_dim, asym);
#else
    const int k = ext_kind(h->cfg);
    if (help) return k == 2 ? tf_occupancy_env_2_2(h->action_dim, asym) : (k == 1 ? tf_occupancy_env_1_2(h->action_dim, asym) : tf_occupancy_env_0_2(h->action_dim, asym));
    if (k == 2) return h->wide ? tf_occupancy_env_2_1(h->action_dim, asym) : tf_occupancy_env_2_0(h->action_dim, asym);
    if (k == 1) return h->wide ? tf_occupancy_env_1_1(h->action_dim, asym) : tf_occupancy_env_1_0(h->action_dim, asym);
    return h->wide ? tf_occupancy_env_0_1(h->action_dim, asym) : tf_occupancy_env_0_0(h->action_dim, asym);
#endif
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

// The fused step kernel k_env<A, IS_RESET, ASYM, MODE, EXT, WIDE> lives in tf_env_kernels.hip, compiled once per (EXT, WIDE) pair so that the six
// translation units build in parallel (tf_launch.h).  The extended domain randomisation (robot base / stage position, per-body friction) and the
// general box object (TfModel.box) are their own instantiations (EXT = 1, 2): the kernels of the headline path stay exactly what they were - a
// run-time flag for either cost 9 to 26 us per step through register pressure - and the extended randomisation with the cube (BASELINE configs[3])
// does not carry the box code either (round 3: with both behind one template flag the cube role of that kernel spilled 95 registers on its serial chain).
static int ext_kind(const TfConfig& c) {
    if (c.model.box) return 2;
    if (!c.dr_enable) return 0;
    for (int i = 0; i < 3; ++i) if (c.dr_base_pos[i] > 0.0f) return 1;
    for (int i = 0; i < 2; ++i) if (c.dr_stage_pos[i] > 0.0f) return 1;
    const float* f[3] = {c.dr_friction_robot, c.dr_friction_object, c.dr_friction_stage};
    for (int b = 0; b < 3; ++b) if (f[b][0] != 1.0f || f[b][1] != 1.0f) return 1;
    return 0;
}
static void launch_env(TfHandle_* h, int lm, const float* action, hipStream_t s) {
    EnvLaunch a;
    a.grid = (unsigned)n_waves(h); a.action_dim = h->action_dim; a.asym = h->cfg.asymmetric_obs != 0;
    a.d_params = h->d_params; a.sa = h->sa; a.action = action; a.stream = s;
    // the helper units carry the launches that simulate; a launch of one of the other hooks (split path) is the plain 256-register kernel
    const bool help = use_helpers(h) && (lm == TF_LM_STEP || lm == TF_LM_STEP_RAND || lm == TF_LM_RESET || lm == TF_LM_SIM);
#if defined(TF_DEV_MIN)      // developer builds (tools/ab_bench.py, tools/variant_sweep.py): the headline kernels only (the fused launches: no split path)
    if (help && lm != TF_LM_SIM) tf_launch_env_0_2(lm, a); else if (h->wide) tf_launch_env_0_1(lm, a); else tf_launch_env_0_0(lm, a);
#else
    const int k = ext_kind(h->cfg);
    if (help) { if (k == 2) tf_launch_env_2_2(lm, a); else if (k == 1) tf_launch_env_1_2(lm, a); else tf_launch_env_0_2(lm, a); }
    else if (k == 2) { if (h->wide) tf_launch_env_2_1(lm, a); else tf_launch_env_2_0(lm, a); }
    else if (k == 1) { if (h->wide) tf_launch_env_1_1(lm, a); else tf_launch_env_1_0(lm, a); }
    else { if (h->wide) tf_launch_env_0_1(lm, a); else tf_launch_env_0_0(lm, a); }
#endif
}

static int launch_step(TfHandle_* h, const float* action, bool is_reset, hipStream_t s, bool random_actions = false) {
    const int nsim = is_reset ? 1 : h->cfg.control_decimation;
    h->frame_count += nsim;
    h->sa.nsim = nsim;
    h->sa.frame = (uint32_t)h->frame_count;
    h->sa.frame0 = (uint32_t)(h->frame_count - nsim);
    reward_coefs(h);
    // a full reset also re-arms the statistics accumulators (they are left at zero by every completed launch; this only
    // matters after a launch that did not complete)
    if (is_reset) HIP_TRY(hipMemsetAsync((void*)h->dp.tickets, 0, STAT_WORDS * sizeof(unsigned long long), s));
    const bool timing = !is_reset && h->ev && h->ev_used < h->ev_cap;
    if (timing && h->ev_phase == 0) HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used], s));      // window opens
    launch_env(h, is_reset ? TF_LM_RESET : (random_actions ? TF_LM_STEP_RAND : TF_LM_STEP), random_actions ? nullptr : action, s);
    LAUNCH_CHECK("k_env");
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
int tf_step_random(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    return launch_step(h, nullptr, false, (hipStream_t)stream, true);
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

// ---- split path: one hook of the reference step per launch (parity tests) ----
int tf_apply_resets(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
#if defined(TF_DEV_MIN)
    return TF_ERR_UNSUPPORTED;
#else
    launch_env(h, TF_LM_RESETS, nullptr, (hipStream_t)stream);
#endif
    LAUNCH_CHECK("k_env<resets>");
    return TF_OK;
}
int tf_pre_step(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    h->sa.frame0 = (uint32_t)h->frame_count;
#if defined(TF_DEV_MIN)
    return TF_ERR_UNSUPPORTED;
#else
    launch_env(h, TF_LM_TORQUE, nullptr, (hipStream_t)stream);
#endif
    LAUNCH_CHECK("k_env<torque>");
    return TF_OK;
}
int tf_simulate(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    h->frame_count += 1;
    h->sa.nsim = 1;
#if defined(TF_DEV_MIN)
    return TF_ERR_UNSUPPORTED;
#else
    launch_env(h, TF_LM_SIM, nullptr, (hipStream_t)stream);
#endif
    LAUNCH_CHECK("k_env<simulate>");
    return TF_OK;
}
int tf_post_step(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
    reward_coefs(h);
    h->sa.frame = (uint32_t)h->frame_count;
#if defined(TF_DEV_MIN)
    return TF_ERR_UNSUPPORTED;
#else
    launch_env(h, TF_LM_POST, nullptr, (hipStream_t)stream);
#endif
    LAUNCH_CHECK("k_env<post>");
    return TF_OK;
}
int tf_finish_step(tf_handle h, void* stream) {
    CHECK_HANDLE(h)
#if defined(TF_DEV_MIN)
    return TF_ERR_UNSUPPORTED;
#else
    launch_env(h, TF_LM_FINISH, nullptr, (hipStream_t)stream);
#endif
    LAUNCH_CHECK("k_env<finish>");
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