// tf_device_math.h - device-side arithmetic shared by every kernel of the TriFinger step: deterministic elementary
// functions, Philox4x32-10, small vector / quaternion helpers, the 3-DoF finger kinematics and dynamics.
//
// Arithmetic contract (shared with the CPU oracle used by the tests): fp32 IEEE add/mul/div/sqrt, explicitly written
// fused multiply-adds and no compiler contraction (-ffp-contract=off), own polynomial sin/cos/exp/asin/log and Newton
// reciprocal / rsqrt, fixed evaluation order.  Per-env outputs are bit-identical to the oracle's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trifinger.h"

#define DEV __device__ __forceinline__

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
DEV void rng4_key(uint32_t k0, uint32_t k1, uint32_t gid, uint32_t count, uint32_t tag, float u[4]) {
    uint32_t r[4];
    philox4x32_10(gid, count, tag, 0u, k0, k1, r);
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

// finger base frame <-> world: world = Rz(yaw) base + (0, 0, H).  The yaw of the finger a wavefront works on is
// wave-uniform (scalar registers).
struct Yaw { float c, s, hc, hs, H; };
DEV void base_to_world(const Yaw& y, const float b[3], float w[3]) {
    w[0] = FMA(y.c, b[0], -(y.s * b[1]));
    w[1] = FMA(y.s, b[0], y.c * b[1]);
    w[2] = b[2] + y.H;
}
DEV void world_to_base(const Yaw& y, const float w[3], float b[3]) {
    b[0] = FMA(y.c, w[0], y.s * w[1]);
    b[1] = FMA(y.c, w[1], -(y.s * w[0]));
    b[2] = w[2] - y.H;
}
DEV void dir_world_to_base(const Yaw& y, const float w[3], float b[3]) {
    b[0] = FMA(y.c, w[0], y.s * w[1]);
    b[1] = FMA(y.c, w[1], -(y.s * w[0]));
    b[2] = w[2];
}
DEV void dir_base_to_world(const Yaw& y, const float b[3], float w[3]) {
    w[0] = FMA(y.c, b[0], -(y.s * b[1]));
    w[1] = FMA(y.s, b[0], y.c * b[1]);
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
