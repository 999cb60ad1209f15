// tf_contact.h - contact geometry and solver rows shared by the finger and cube roles (tf_roles.h).
// Every function is the arithmetic twin of the function of the same name in the test oracle (oracle/tf_oracle.c):
// same operations in the same order, so that per-env results agree bit for bit.
#pragma once
#include "tf_params.h"

// ---- PGS row kernels ----
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

// A contact slot is LIVE (gets rows) when its gap is inside the broad margin and can close within this substep at the approach
// speed of the free velocities, plus a slack for what other impulses may add.  Everything a dead slot would do is skipped:
// the rows sit behind per-lane branches, so a wavefront with no live lane for a slot jumps over them.
DEV bool contact_live(const TfModel& m, float gap, float vn0, float h) {
    return (gap < m.contact_margin) && (gap < FMA(h, f_max(-vn0, 0.0f), m.contact_slack));
}

// base-frame position of a point given in the frame of link LINK (1..3)
template <int LINK> DEV void link_point(const FK& k, const float local[3], float out[3]) {
    float t[3];
    rot_link<LINK>(k, local, t);
    if (LINK == 1) { out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; }
    else if (LINK == 2) { out[0] = k.p2[0] + t[0]; out[1] = k.p2[1] + t[1]; out[2] = k.p2[2] + t[2]; }
    else { out[0] = k.p3[0] + t[0]; out[1] = k.p3[1] + t[1]; out[2] = k.p3[2] + t[2]; }
}

// Joint-space rows of a contact on link `link` (per-lane value) at base-frame point Pb for the three world directions
// dirs[3d..3d+2]: J[3d..] = (L1.d, L2.d, L3.d) with the levers of the joints that do not move the link zeroed,
// W[3d..] = M^-1 J, Dd[d] = J.W.
DEV void finger_jac(const Yaw& y, const FK& k, int link, const float Pb[3], const float dirs[9], float J[9], float W[9], float Dd[3]) {
    float L1[3], L2[3], L3[3];
    levers(k, Pb, L1, L2, L3);
    if (link < 2) { L2[0] = 0.0f; L2[1] = 0.0f; L2[2] = 0.0f; }
    if (link < 3) { L3[0] = 0.0f; L3[1] = 0.0f; L3[2] = 0.0f; }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float db[3];
        dir_world_to_base(y, &dirs[3 * d], db);
        J[3 * d] = dot3(L1, db); J[3 * d + 1] = dot3(L2, db); J[3 * d + 2] = dot3(L3, db);
        sym3_mul(k.Minv, &J[3 * d], &W[3 * d]);
        Dd[d] = dot3(&J[3 * d], &W[3 * d]);
    }
}

// g(s) = d . (x - clamp(x)) with x = a + s d: half the derivative of the squared distance between the segment point x(s)
// and the box [-hc, hc]^3; monotone non-decreasing and piecewise linear in s
DEV float seg_box_g(const float a[3], const float d[3], float s, const float hc[3]) {
    float e[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { float x = FMA(s, d[i], a[i]); e[i] = x - f_clamp(x, -hc[i], hc[i]); }
    return dot3(d, e);
}
// Closest points between the segment a + s (b - a) and the box [-hc, hc]^3 (box frame), exact: g changes slope only where
// a coordinate of x(s) crosses +-hc (at most six breakpoints), so its root lies on the straight piece between the last
// breakpoint with g <= 0 and the first with g > 0 (end points included): one linear interpolation, no iteration, no
// branches.  x on the segment, y on the box, unit direction nc from y to x, gap = |x - y| - radius.  A segment point inside
// the box is pushed out through the nearest face.
DEV void seg_box(const float a[3], const float b[3], const float hc[3], float radius, float& gap_out, float x[3], float y[3], float nc[3], float& s_out) {
    float d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const float g0 = seg_box_g(a, d, 0.0f, hc), g1 = seg_box_g(a, d, 1.0f, hc);
    float lo = 0.0f, glo = g0, hi = 1.0f, ghi = g1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float ad = f_abs(d[i]);
        const bool ok = ad > 1e-9f;
        float inv = ok ? f_rcp(ok ? ad : 1.0f) : 0.0f;
        inv = (d[i] < 0.0f) ? -inv : inv;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const float sb = ((side ? hc[i] : -hc[i]) - a[i]) * inv;
            const float gb = seg_box_g(a, d, sb, hc);
            const bool valid = ok && sb > 0.0f && sb < 1.0f;
            const bool take_lo = valid && gb <= 0.0f && sb > lo;
            lo = take_lo ? sb : lo; glo = take_lo ? gb : glo;
            const bool take_hi = valid && gb > 0.0f && sb < hi;
            hi = take_hi ? sb : hi; ghi = take_hi ? gb : ghi;
        }
    }
    float s = f_clamp(FMA(-glo, (hi - lo) * f_rcp(f_max(ghi - glo, 1e-30f)), lo), lo, hi);
    if (g0 > 0.0f) s = 0.0f;
    if (!(g1 > 0.0f)) s = 1.0f;
    s_out = s;
#pragma unroll
    for (int i = 0; i < 3; ++i) { x[i] = FMA(s, d[i], a[i]); y[i] = f_clamp(x[i], -hc[i], hc[i]); }
    float ev[3] = {x[0] - y[0], x[1] - y[1], x[2] - y[2]};
    float dist2 = dot3(ev, ev);
    if (__builtin_expect(dist2 > 1e-12f, 1)) {
        float inv = f_rsqrt(dist2);
        float dist = dist2 * inv;
        nc[0] = ev[0] * inv; nc[1] = ev[1] * inv; nc[2] = ev[2] * inv;
        gap_out = dist - radius;
    } else {
        int bi = 0;
        float best = f_abs(x[0]) - hc[0];
        float p1 = f_abs(x[1]) - hc[1];
        if (p1 > best) { best = p1; bi = 1; }
        float p2 = f_abs(x[2]) - hc[2];
        if (p2 > best) { best = p2; bi = 2; }
        float xb = (bi == 0) ? x[0] : ((bi == 1) ? x[1] : x[2]);
        float sg = (xb < 0.0f) ? -1.0f : 1.0f;
        nc[0] = (bi == 0) ? sg : 0.0f; nc[1] = (bi == 1) ? sg : 0.0f; nc[2] = (bi == 2) ? sg : 0.0f;
        y[0] = (bi == 0) ? sg * hc[0] : y[0]; y[1] = (bi == 1) ? sg * hc[1] : y[1]; y[2] = (bi == 2) ? sg * hc[2] : y[2];
        gap_out = best - radius;
    }
}

// the same for a sphere (centre x in the box frame): the tail of seg_box for a segment of zero length
// (x, y, nc, gap as in seg_box)
DEV void point_box(const float x[3], const float hc[3], float radius, float& gap_out, float y[3], float nc[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = f_clamp(x[i], -hc[i], hc[i]);
    float ev[3] = {x[0] - y[0], x[1] - y[1], x[2] - y[2]};
    float dist2 = dot3(ev, ev);
    if (__builtin_expect(dist2 > 1e-12f, 1)) {
        float inv = f_rsqrt(dist2);
        float dist = dist2 * inv;
        nc[0] = ev[0] * inv; nc[1] = ev[1] * inv; nc[2] = ev[2] * inv;
        gap_out = dist - radius;
    } else {
        int bi = 0;
        float best = f_abs(x[0]) - hc[0];
        float p1 = f_abs(x[1]) - hc[1];
        if (p1 > best) { best = p1; bi = 1; }
        float p2 = f_abs(x[2]) - hc[2];
        if (p2 > best) { best = p2; bi = 2; }
        float xb = (bi == 0) ? x[0] : ((bi == 1) ? x[1] : x[2]);
        float sg = (xb < 0.0f) ? -1.0f : 1.0f;
        nc[0] = (bi == 0) ? sg : 0.0f; nc[1] = (bi == 1) ? sg : 0.0f; nc[2] = (bi == 2) ? sg : 0.0f;
        y[0] = (bi == 0) ? sg * hc[0] : y[0]; y[1] = (bi == 1) ? sg * hc[1] : y[1]; y[2] = (bi == 2) ? sg * hc[2] : y[2];
        gap_out = best - radius;
    }
}

// closest points of two segments p1-q1 and p2-q2 (Ericson, Real-Time Collision Detection 5.1.9; both of positive length)
DEV void seg_seg(const float p1[3], const float q1[3], const float p2[3], const float q2[3], float c1[3], float c2[3]) {
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
#pragma unroll
    for (int i = 0; i < 3; ++i) { c1[i] = FMA(s, d1[i], p1[i]); c2[i] = FMA(t, d2[i], p2[i]); }
}

// the same with the parameter s of the closest point on the first segment (the cross-section of a link shape depends on it)
DEV void seg_seg_s(const float p1[3], const float q1[3], const float p2[3], const float q2[3], float c1[3], float c2[3], float& s_out) {
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
#pragma unroll
    for (int i = 0; i < 3; ++i) { c1[i] = FMA(s, d1[i], p1[i]); c2[i] = FMA(t, d2[i], p2[i]); }
    s_out = s;
}

// inner radius of the boundary at height z: the piecewise-linear profile through the knots (wall_z[i], wall_r[i]) - a vertical ring below
// the first knot, the flaring cone of the stage above it (slopes wall_s precomputed at tf_create), nothing above the last knot (1e3)
DEV float wall_radius_at(const DevParams& P, float z) {
    const TfModel& m = P.m;
    float r = m.wall_r[0];
    r = (z > m.wall_z[0]) ? FMA(z - m.wall_z[0], P.wall_s[0], m.wall_r[0]) : r;
    r = (z > m.wall_z[1]) ? FMA(z - m.wall_z[1], P.wall_s[1], m.wall_r[1]) : r;
    r = (z > m.wall_z[2]) ? FMA(z - m.wall_z[2], P.wall_s[2], m.wall_r[2]) : r;
    r = (z < m.wall_z[3]) ? r : 1000.0f;
    return r;
}

// the same with the TILT of the surface at that height: (c, sn) = (cos, sin) of the slope angle of the profile segment, (1, 0) on the vertical ring.
// The inward surface normal is (c n_h, sn) with n_h the inward horizontal unit vector (the stage is a bowl: above 32 mm its wall leans outward by
// 29-35 degrees, high_table_boundary.urdf:20-259); the distance of a point at radius rho to the surface is (r(z) - rho) c.  Used by the fingertip -
// boundary contact; the cube corners keep the horizontal normal (their tilted rows cost the cube wavefront 5 us: DESIGN.md section 4).
DEV float wall_profile(const DevParams& P, float z, float& c, float& sn) {
    const TfModel& m = P.m;
    const bool b0 = z > m.wall_z[0], b1 = z > m.wall_z[1], b2 = z > m.wall_z[2];
    float r = m.wall_r[0];
    c = 1.0f; sn = 0.0f;
    r = b0 ? FMA(z - m.wall_z[0], P.wall_s[0], m.wall_r[0]) : r;  c = b0 ? P.wall_c[0] : c;  sn = b0 ? P.wall_sn[0] : sn;
    r = b1 ? FMA(z - m.wall_z[1], P.wall_s[1], m.wall_r[1]) : r;  c = b1 ? P.wall_c[1] : c;  sn = b1 ? P.wall_sn[1] : sn;
    r = b2 ? FMA(z - m.wall_z[2], P.wall_s[2], m.wall_r[2]) : r;  c = b2 ? P.wall_c[2] : c;  sn = b2 ? P.wall_sn[2] : sn;
    r = (z < m.wall_z[3]) ? r : 1000.0f;
    return r;
}

DEV void cube_corner(const float R[9], const float hc[3], int k, float sk, int idx, float r[3]) {
    // axes a < b are the two that are not k: bit 0 of idx is the sign along a, bit 1 the sign along b
    float y[3];
    const bool pa = (idx & 1) != 0, pb = (idx & 2) != 0;
    y[0] = (k == 0) ? sk * hc[0] : (pa ? hc[0] : -hc[0]);
    y[1] = (k == 1) ? sk * hc[1] : (((k == 0) ? pa : pb) ? hc[1] : -hc[1]);
    y[2] = (k == 2) ? sk * hc[2] : (pb ? hc[2] : -hc[2]);
    mat3_mul(R, y, r);
}

// ---- general box (TfModel.box): rows with explicit arms in INERTIA-SCALED angular coordinates (oracle: same names).  With
// S = R diag(sqrt(I_ref / I_k)) R^T the substitution w = S w^, a^ = S a keeps every row in the isotropic form with
// inv_I = 1 / I_ref; only its arm is S (r x n) instead of r x n.
DEV void box_arm(const float S[6], const float r[3], const float n[3], float a[3]) {
    float c[3];
    cross3(r, n, c);
    sym3_mul(S, c, a);
}
DEV float g_vrel(const float n[3], const float a[3], const float v[3], const float w[3]) { return dot3(n, v) + dot3(a, w); }
DEV void g_apply(const float n[3], const float a[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
#pragma unroll
    for (int j = 0; j < 3; ++j) { v[j] = FMA(n[j], s, v[j]); w[j] = FMA(a[j], q, w[j]); }
}
DEV void rot_diag_rot(const float R[9], const float s[3], float S[6]) {     // R diag(s) R^T as 00 01 02 11 12 22
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int i = (e < 3) ? 0 : ((e < 5) ? 1 : 2), j = (e < 3) ? e : ((e < 5) ? e - 2 : 2);
        S[e] = FMA(R[3 * i + 2] * s[2], R[3 * j + 2], FMA(R[3 * i + 1] * s[1], R[3 * j + 1], (R[3 * i] * s[0]) * R[3 * j]));
    }
}
DEV void box_axis(int d, float n[3]) {        // rows +z, +x, +y of a floor corner
    n[0] = (d == 1) ? 1.0f : 0.0f; n[1] = (d == 2) ? 1.0f : 0.0f; n[2] = (d == 0) ? 1.0f : 0.0f;
}

// axis-aligned rows of a cube corner with arm r: direction +z / +x / +y.  *_vrel: relative velocity of the row, *_apply:
// effect of the impulse dl on the cube
DEV float cz_vrel(const float r[3], const float v[3], const float w[3]) { return FMA(r[1], w[0], FMA(-r[0], w[1], v[2])); }
DEV void cz_apply(const float r[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[2] = v[2] + s;
    w[0] = FMA(r[1], q, w[0]);
    w[1] = FMA(-r[0], q, w[1]);
}
DEV float cx_vrel(const float r[3], const float v[3], const float w[3]) { return FMA(r[2], w[1], FMA(-r[1], w[2], v[0])); }
DEV void cx_apply(const float r[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = v[0] + s;
    w[1] = FMA(r[2], q, w[1]);
    w[2] = FMA(-r[1], q, w[2]);
}
DEV float cy_vrel(const float r[3], const float v[3], const float w[3]) { return FMA(-r[2], w[0], FMA(r[0], w[2], v[1])); }
DEV void cy_apply(const float r[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[1] = v[1] + s;
    w[0] = FMA(-r[2], q, w[0]);
    w[2] = FMA(r[0], q, w[2]);
}
// wall rows: inward horizontal normal n = (n0, n1, 0) and tangent t = (-n1, n0, 0)
DEV void wall_arm_n(const float r[3], const float n[2], float a[3]) {
    a[0] = -(r[2] * n[1]);
    a[1] = r[2] * n[0];
    a[2] = FMA(r[0], n[1], -(r[1] * n[0]));
}
DEV void wall_arm_t(const float r[3], const float n[2], float b[3]) {
    b[0] = -(r[2] * n[0]);
    b[1] = -(r[2] * n[1]);
    b[2] = FMA(r[0], n[0], r[1] * n[1]);
}
DEV float wn_vrel(const float n[2], const float a[3], const float v[3], const float w[3]) {
    return FMA(a[2], w[2], FMA(a[1], w[1], FMA(a[0], w[0], FMA(n[1], v[1], n[0] * v[0]))));
}
DEV void wn_apply(const float n[2], const float a[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = FMA(n[0], s, v[0]);
    v[1] = FMA(n[1], s, v[1]);
    w[0] = FMA(a[0], q, w[0]); w[1] = FMA(a[1], q, w[1]); w[2] = FMA(a[2], q, w[2]);
}
DEV float wt_vrel(const float n[2], const float b[3], const float v[3], const float w[3]) {
    return FMA(b[2], w[2], FMA(b[1], w[1], FMA(b[0], w[0], FMA(n[0], v[1], -(n[1] * v[0])))));
}
DEV void wt_apply(const float n[2], const float b[3], float dl, float inv_m, float inv_I, float v[3], float w[3]) {
    float s = dl * inv_m, q = dl * inv_I;
    v[0] = FMA(-n[1], s, v[0]);
    v[1] = FMA(n[0], s, v[1]);
    w[0] = FMA(b[0], q, w[0]); w[1] = FMA(b[1], q, w[1]); w[2] = FMA(b[2], q, w[2]);
}

// ---- the same rows on a PACKED twist: p[j] = (v_j, w_j) as a register pair, so that the general rows of the cube role's 256-register instantiation
// run as v_pk_mul_f32 / v_pk_fma_f32 (a lone wavefront issues a packed fp32 instruction in the same ~5 cycles as a scalar one: tools/microbench/
// valu_pk.hip - with one wavefront per SIMD the step is bound by issue slots, not by the vector ALU).  Every lane-operation is the one the scalar
// form performs, in the same order: bit-identical results.
typedef float float2v __attribute__((ext_vector_type(2)));
struct Twist { float2v p[3]; };
DEV float2v pk_fma(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }
DEV float2v pk_splat(float x) { float2v r = {x, x}; return r; }
// dot3(dir, v) + dot3(rxd, w) with rec[j] = (dir_j, rxd_j)
DEV float pk_row_vel(const float2v rec[3], const Twist& t) {
    float2v p = rec[0] * t.p[0];
    p = pk_fma(rec[1], t.p[1], p);
    p = pk_fma(rec[2], t.p[2], p);
    return p.x + p.y;
}
// v -= dir dl / m, w -= rxd dl / I with mI = (1/m, 1/I)
DEV void pk_row_apply_neg(const float2v rec[3], float dl, float2v mI, Twist& t) {
    const float2v s = pk_splat(dl) * mI;
#pragma unroll
    for (int j = 0; j < 3; ++j) t.p[j] = pk_fma(-rec[j], s, t.p[j]);
}
DEV float cz_vrel(const float r[3], const Twist& t) { return FMA(r[1], t.p[0].y, FMA(-r[0], t.p[1].y, t.p[2].x)); }
DEV void cz_apply(const float r[3], float dl, float2v mI, Twist& t) {
    const float2v s = pk_splat(dl) * mI;
    t.p[2].x = t.p[2].x + s.x;
    t.p[0].y = FMA(r[1], s.y, t.p[0].y);
    t.p[1].y = FMA(-r[0], s.y, t.p[1].y);
}
DEV float cx_vrel(const float r[3], const Twist& t) { return FMA(r[2], t.p[1].y, FMA(-r[1], t.p[2].y, t.p[0].x)); }
DEV void cx_apply(const float r[3], float dl, float2v mI, Twist& t) {
    const float2v s = pk_splat(dl) * mI;
    t.p[0].x = t.p[0].x + s.x;
    t.p[1].y = FMA(r[2], s.y, t.p[1].y);
    t.p[2].y = FMA(-r[1], s.y, t.p[2].y);
}
DEV float cy_vrel(const float r[3], const Twist& t) { return FMA(-r[2], t.p[0].y, FMA(r[0], t.p[2].y, t.p[1].x)); }
DEV void cy_apply(const float r[3], float dl, float2v mI, Twist& t) {
    const float2v s = pk_splat(dl) * mI;
    t.p[1].x = t.p[1].x + s.x;
    t.p[0].y = FMA(-r[2], s.y, t.p[0].y);
    t.p[2].y = FMA(r[0], s.y, t.p[2].y);
}
DEV float wn_vrel(const float n[2], const float a[3], const Twist& t) {
    return FMA(a[2], t.p[2].y, FMA(a[1], t.p[1].y, FMA(a[0], t.p[0].y, FMA(n[1], t.p[1].x, n[0] * t.p[0].x))));
}
DEV void wn_apply(const float n[2], const float a[3], float dl, float2v mI, Twist& t) {
    const float2v s = pk_splat(dl) * mI;
    t.p[0].x = FMA(n[0], s.x, t.p[0].x);
    t.p[1].x = FMA(n[1], s.x, t.p[1].x);
    t.p[0].y = FMA(a[0], s.y, t.p[0].y); t.p[1].y = FMA(a[1], s.y, t.p[1].y); t.p[2].y = FMA(a[2], s.y, t.p[2].y);
}
DEV float wt_vrel(const float n[2], const float b[3], const Twist& t) {
    return FMA(b[2], t.p[2].y, FMA(b[1], t.p[1].y, FMA(b[0], t.p[0].y, FMA(n[0], t.p[1].x, -(n[1] * t.p[0].x)))));
}
DEV void wt_apply(const float n[2], const float b[3], float dl, float2v mI, Twist& t) {
    const float2v s = pk_splat(dl) * mI;
    t.p[0].x = FMA(-n[1], s.x, t.p[0].x);
    t.p[1].x = FMA(n[0], s.x, t.p[1].x);
    t.p[0].y = FMA(b[0], s.y, t.p[0].y); t.p[1].y = FMA(b[1], s.y, t.p[1].y); t.p[2].y = FMA(b[2], s.y, t.p[2].y);
}
