"""Developer experiment (CPU, fp64), round 4: BLOCK variants of the projected Gauss-Seidel sweep on the contact problems dumped by
tests/dev/pgs_cases.py - which part of the problem do 8 sweeps fail to resolve, and what would an exact solve of that part buy?
("finger-cube blocks on the finger wavefronts in parallel and/or a direct 6x6 cube solve".)

  exact sub-blocks      per outer sweep one block of rows is iterated to convergence (50 inner sweeps), the rest swept once:
                        floor / wall rows; floor / wall normals only; ALL rows that touch the cube (finger-cube + floor + wall);
                        the finger-only rows (fingertip-floor / -wall, joint limits); finger-cube + finger-only rows
  cheap extras          k normal-only or cube-block sweeps appended to every full sweep
  overlapping blocks    [finger f: its finger-cube + finger-only rows, Kf inner sweeps, the three fingers in parallel from one snapshot of the
                        cube twist] then [rows touching the cube, Kc inner sweeps]
  parallel fingers      the review's variant (a): the finger-cube rows solved by the three fingers in parallel (Jacobi across fingers, cube
                        inverse inertia x number of live finger contacts = mass splitting), then floor / wall rows
  composite body        floor / wall rows solved on the cube WITH the sticking finger contacts attached through K = A^-1 (Schur complement)

   python tests/dev/pgs_cases.py 128 /tmp/pgs_cases.pkl && python tests/dev/pgs_blocks.py /tmp/pgs_cases.pkl      -> profiles/r4_a_solver_variants.txt"""
import sys, pickle
import numpy as np
from pgs_accel import Prob, plain, report


def classify(P):
    fc, fing, cube = [], [], []
    for i in range(P.n):
        if P.D[i] <= 0: continue
        tj, tc = P.touch_joint[i], P.touch_cube[i]
        (fc if (tj and tc) else (fing if tj else cube)).append(i)
    return fc, fing, cube


def exact_block(P, k, which, inner=50):
    fc, fing, cube = classify(P)
    lam = np.zeros(P.n); v = P.v0.copy()
    cn = [i for i in cube if P.kind[i] == 0]
    for _ in range(k):
        if which == 'floor':
            P.sweep(lam, v, fc); P.sweep(lam, v, fing)
            for _ in range(inner): P.sweep(lam, v, cube)
        elif which == 'floor normals':
            P.sweep(lam, v, fc); P.sweep(lam, v, fing)
            for _ in range(inner): P.sweep(lam, v, cn)
            P.sweep(lam, v, cube)
        elif which == 'cube block':
            for _ in range(inner): P.sweep(lam, v, fc + cube)
            P.sweep(lam, v, fing)
        elif which == 'finger-only':
            P.sweep(lam, v, fc)
            for _ in range(inner): P.sweep(lam, v, fing)
            P.sweep(lam, v, cube)
        elif which == 'fc + finger-only':
            for _ in range(inner): P.sweep(lam, v, fc + fing)
            P.sweep(lam, v, cube)
    return v


def extras(P, k, extra_normals=0, extra_cube=0):
    fc, fing, cube = classify(P)
    blk = fc + cube
    nrm = [i for i in blk if P.kind[i] == 0]
    lam = np.zeros(P.n); v = P.v0.copy()
    for _ in range(k):
        P.sweep(lam, v, fc); P.sweep(lam, v, fing); P.sweep(lam, v, cube)
        for _ in range(extra_normals): P.sweep(lam, v, nrm)
        for _ in range(extra_cube): P.sweep(lam, v, blk)
    return v


def finger_of(P, i): return int(np.argmax(np.abs(P.J[i][:9]) > 0)) // 3


def overlapping(P, outer, Kf, Kc):
    fc, fing, cube = classify(P)
    byf = [[i for i in fc + fing if finger_of(P, i) == f] for f in range(3)]
    blk = fc + cube
    lam = np.zeros(P.n); v = P.v0.copy()
    for _ in range(outer):
        snap = v.copy(); tot = np.zeros(15)
        for f in range(3):
            vv = snap.copy()
            for _ in range(Kf): P.sweep(lam, vv, byf[f])
            tot += vv - snap
        v = snap + tot
        for _ in range(Kc): P.sweep(lam, v, blk)
    return v


def parallel_fingers(P, k, msplit=True):
    fc, fing, cube = classify(P)
    byf = [[i for i in fc + fing if finger_of(P, i) == f] for f in range(3)]
    nc = sum(1 for f in range(3) if any(i in fc for i in byf[f]))
    s = float(max(nc, 1)) if msplit else 1.0
    Minv2 = P.Minv.copy(); Minv2[9:, 9:] *= s
    W2 = P.J @ Minv2; D2 = np.einsum('ij,ij->i', P.J, W2)
    lam = np.zeros(P.n); v = P.v0.copy()
    for _ in range(k):
        snap = v.copy()
        for f in range(3):
            vv = snap.copy()
            for i in byf[f]:
                d = D2[i]; vr = P.J[i] @ vv
                if P.kind[i] == 0: new = max(lam[i] - (vr + P.bias[i]) / d, 0.0)
                elif P.kind[i] == 1:
                    lim = P.mu[i] * lam[P.parent[i]]; new = min(max(lam[i] - vr / d, -lim), lim)
                else:
                    v0_ = vr - d * lam[i]; new = (min(max(v0_, P.lo[i]), P.hi[i]) - v0_) / d
                dl = new - lam[i]; lam[i] = new; vv += W2[i] * dl
        v = P.v0 + lam @ P.W
        P.sweep(lam, v, cube)
    return v


def composite(P, k):
    fc, fing, cube = classify(P)
    lam = np.zeros(P.n); v = P.v0.copy()
    Mc = np.linalg.inv(P.Minv[9:, 9:])
    trip = [fc[i:i + 3] for i in range(0, len(fc), 3)]
    for _ in range(k):
        P.sweep(lam, v, fc); P.sweep(lam, v, fing)
        H = Mc.copy(); inc = []
        for n, t1, t2 in trip:
            rows = []
            if lam[n] > 0:
                rows.append(n)
                rows += [tt for tt in (t1, t2) if abs(lam[tt]) < P.mu[tt] * lam[n]]
            if not rows: continue
            Jf, G = P.J[rows][:, :9], P.J[rows][:, 9:]
            A = Jf @ P.Minv[:9, :9] @ Jf.T
            K = np.linalg.inv(A + 1e-4 * np.trace(A) * np.eye(len(rows)))
            H += G.T @ K @ G; inc.append((rows, K, G))
        Hinv = np.linalg.inv(H)
        x0 = v[9:].copy(); x = x0.copy()
        for i in cube:
            g = P.J[i][9:]; w = Hinv @ g; d = g @ w; vr = g @ x
            if P.kind[i] == 0: new = max(lam[i] - (vr + P.bias[i]) / d, 0.0)
            else:
                lim = P.mu[i] * lam[P.parent[i]]; new = min(max(lam[i] - vr / d, -lim), lim)
            x = x + w * (new - lam[i]); lam[i] = new
        for rows, K, G in inc:
            for r, dlr in zip(rows, -K @ (G @ (x - x0))): lam[r] += dlr
        for n, t1, t2 in trip:
            lam[n] = max(lam[n], 0.0)
            for tt in (t1, t2):
                lim = P.mu[tt] * lam[n]; lam[tt] = min(max(lam[tt], -lim), lim)
        v = P.v0 + lam @ P.W
    return v


if __name__ == '__main__':
    cases = pickle.load(open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/pgs_cases.pkl', 'rb'))
    Ps = [Prob(c) for c in cases]
    print(len(Ps), 'cases (cold start, error against the fp64 fixed point, scaled as in tests/test_contact_lcp_reference.py)')
    for k in (8, 12, 16): report(f'plain PGS (the spec), {k}', [P.err(plain(P, k)) for P in Ps], cases)
    for k in (4, 8):
        for w in ('floor', 'floor normals', 'finger-only', 'fc + finger-only', 'cube block'):
            report(f'exact block [{w}], outer {k}', [P.err(exact_block(P, k, w)) for P in Ps], cases)
    for k, en, ec in ((8, 1, 0), (8, 3, 0), (8, 0, 1), (6, 0, 1), (4, 0, 3)):
        report(f'outer {k} + {en} normal-only + {ec} cube-block extras', [P.err(extras(P, k, en, ec)) for P in Ps], cases)
    for o, kf, kc in ((8, 1, 1), (4, 4, 2), (8, 2, 2), (8, 4, 2), (4, 8, 4), (8, 1, 4)):
        report(f'overlapping blocks outer {o} Kf {kf} Kc {kc}', [P.err(overlapping(P, o, kf, kc)) for P in Ps], cases)
    for k in (8, 16):
        report(f'parallel fingers, mass splitting, {k}', [P.err(parallel_fingers(P, k)) for P in Ps], cases)
        report(f'parallel fingers, no splitting, {k}', [P.err(parallel_fingers(P, k, False)) for P in Ps], cases)
    multi = [i for i, c in enumerate(cases) if c['nfc'] >= 2]
    print('   cases with two or three finger contacts:', len(multi))
    for k in (8, 16):
        print(f'     {k:2d} sweeps  plain           ', ' '.join('%.1e' % Ps[i].err(plain(Ps[i], k)) for i in multi))
        print(f'     {k:2d} sweeps  parallel, split ', ' '.join('%.1e' % Ps[i].err(parallel_fingers(Ps[i], k)) for i in multi))
        print(f'     {k:2d} sweeps  parallel, plain ', ' '.join('%.1e' % Ps[i].err(parallel_fingers(Ps[i], k, False)) for i in multi))
    for k in (4, 8): report(f'composite body for the floor / wall rows, {k}', [P.err(composite(P, k)) for P in Ps], cases)
