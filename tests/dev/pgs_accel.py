"""Developer experiment (CPU, fp64): ACCELERATED variants of the projected Gauss-Seidel sweep on the contact problems dumped by
tests/dev/pgs_cases.py (cut the warm-state residual of a fixed small number of sweeps).

Variants: plain PGS; whole-sweep momentum (Nesterov sequence, fixed beta, with / without restart); Anderson acceleration (depth 1, 2);
nonsmooth nonlinear CG (Silcowitz-Hansen et al. 2010); block Gauss-Seidel with the rows that touch the cube iterated K times per outer sweep.
Error = scaled velocity error against the fp64 fixed point of the problem.   python tests/dev/pgs_accel.py [cases.pkl]"""
import sys, pickle
import numpy as np


class Prob:
    def __init__(self, c):
        r = c['rows']
        self.J, self.kind, self.bias, self.mu, self.lo, self.hi, self.parent = r['J'], r['kind'], r['bias'], r['mu'], r['lo'], r['hi'], r['parent']
        self.Minv, self.v0, self.vstar = c['Minv'], c['v0'], c['vstar']
        self.W = self.J @ self.Minv            # rows: M^-1 J_i^T
        self.D = np.einsum('ij,ij->i', self.J, self.W)
        self.n = len(self.D)
        self.touch_cube = np.any(self.J[:, 9:] != 0, axis=1)
        self.touch_joint = np.any(self.J[:, :9] != 0, axis=1)

    def row(self, i, lam, v):
        d = self.D[i]
        if d <= 0: return 0.0
        vr = self.J[i] @ v
        k = self.kind[i]
        if k == 0: new = max(lam[i] - (vr + self.bias[i]) / d, 0.0)
        elif k == 1:
            lim = self.mu[i] * lam[self.parent[i]]; new = min(max(lam[i] - vr / d, -lim), lim)
        else:
            v0_ = vr - d * lam[i]; new = (min(max(v0_, self.lo[i]), self.hi[i]) - v0_) / d
        dl = new - lam[i]
        if dl != 0.0:
            v += self.W[i] * dl; lam[i] = new
        return dl

    def sweep(self, lam, v, idx=None):
        for i in (range(self.n) if idx is None else idx): self.row(i, lam, v)

    def project(self, lam):
        """feasible set of the rows given the normals (limit rows are free)"""
        for i in range(self.n):
            if self.kind[i] == 0: lam[i] = max(lam[i], 0.0)
        for i in range(self.n):
            if self.kind[i] == 1:
                lim = self.mu[i] * lam[self.parent[i]]; lam[i] = min(max(lam[i], -lim), lim)
        return lam

    def vel(self, lam): return self.v0 + lam @ self.W

    def err(self, v):
        vs = self.vstar
        return max(np.abs(v[:9] - vs[:9]).max() / 10, np.abs(v[9:12] - vs[9:12]).max(), np.abs(v[12:15] - vs[12:15]).max() / 20)


def plain(P, k):
    lam = np.zeros(P.n); v = P.v0.copy()
    for _ in range(k): P.sweep(lam, v)
    return v


def momentum(P, k, beta_fn, restart=False, proj=False, final_plain=1):
    """y_{j+1} = x_{j+1} + beta_j (x_{j+1} - x_j), x_{j+1} = GS(y_j); the last `final_plain` sweeps are not extrapolated"""
    lam = np.zeros(P.n); v = P.v0.copy()
    x_prev = lam.copy(); v_prev = v.copy(); t = 0
    for j in range(k):
        y_lam = lam.copy()
        P.sweep(lam, v)
        x = lam.copy(); xv = v.copy()
        if j < k - final_plain:
            b = beta_fn(t)
            if restart and j > 0 and np.dot(x - y_lam, x - x_prev) < 0: t = 0; b = 0.0
            else: t += 1
            lam = x + b * (x - x_prev)
            if proj: lam = P.project(lam); v = P.vel(lam)
            else: v = xv + b * (xv - v_prev)
        x_prev, v_prev = x, xv
    return v


def sweep_relax(P, k, omega, final_plain=1):
    """y_{j+1} = y_j + omega (GS(y_j) - y_j): relaxation of the whole sweep operator"""
    lam = np.zeros(P.n); v = P.v0.copy()
    for j in range(k):
        l0 = lam.copy(); v0 = v.copy()
        P.sweep(lam, v)
        if j < k - final_plain:
            lam = l0 + omega * (lam - l0); v = v0 + omega * (v - v0)
    return v


def anderson(P, k, m=1, final_plain=1, proj=True):
    lam = np.zeros(P.n); v = P.v0.copy()
    Xs, Gs = [], []          # iterates y_j and their images g_j = GS(y_j)
    for j in range(k):
        y = lam.copy()
        P.sweep(lam, v)
        g = lam.copy()
        Xs.append(y); Gs.append(g)
        if j < k - final_plain and len(Xs) >= 2:
            mm = min(m, len(Xs) - 1)
            F = [Gs[-1 - i] - Xs[-1 - i] for i in range(mm + 1)]      # residuals
            dF = np.array([F[0] - F[i + 1] for i in range(mm)]).T
            try: gamma = np.linalg.lstsq(dF, F[0], rcond=None)[0]
            except Exception: gamma = np.zeros(mm)
            new = Gs[-1].copy()
            for i in range(mm): new -= gamma[i] * (Gs[-1] - Gs[-2 - i])
            lam = P.project(new) if proj else new
            v = P.vel(lam)
    return v


def nncg(P, k, final_plain=0):
    lam = np.zeros(P.n); v = P.v0.copy()
    l0 = lam.copy(); P.sweep(lam, v)
    g = -(lam - l0); p = -g; gg_prev = g @ g
    for j in range(1, k):
        l0 = lam.copy(); P.sweep(lam, v)
        g = -(lam - l0); gg = g @ g
        if j >= k - final_plain: continue
        beta = gg / gg_prev if gg_prev > 0 else 0.0
        if beta > 1.0: p = np.zeros(P.n)
        else:
            lam = lam + beta * p; v = v + (beta * p) @ P.W
            p = beta * p - g
        gg_prev = gg
    return v


def cube_block(P, k_outer, k_inner, accel=None):
    """outer sweep = [rows touching the cube, k_inner times] then [finger-only rows]"""
    lam = np.zeros(P.n); v = P.v0.copy()
    ci = [i for i in range(P.n) if P.touch_cube[i]]; fi = [i for i in range(P.n) if not P.touch_cube[i]]
    for _ in range(k_outer):
        for _ in range(k_inner): P.sweep(lam, v, ci)
        P.sweep(lam, v, fi)
    return v


def report(name, errs, cases):
    e = np.array(errs)
    ch = np.array([c['nfloor'] == 4 and c['nfc'] >= 1 for c in cases])
    s = f"{name:44s} median {np.median(e):.2e}  p90 {np.percentile(e, 90):.2e}  p99 {np.percentile(e, 99):.2e}  max {e.max():.2e}"
    if ch.any(): s += f"   chains: median {np.median(e[ch]):.2e} p90 {np.percentile(e[ch], 90):.2e}"
    print(s, flush=True)


if __name__ == '__main__':
    cases = pickle.load(open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/pgs_cases.pkl', 'rb'))
    Ps = [Prob(c) for c in cases]
    print(len(Ps), 'cases; rows per case: median', int(np.median([p.n for p in Ps])))
    nest = lambda t: t / (t + 3.0)
    V = []
    for k in (8, 12, 16): V.append((f'plain PGS, {k}', lambda P, k=k: plain(P, k)))
    for k in (6, 8):
        V.append((f'Nesterov t/(t+3), {k}', lambda P, k=k: momentum(P, k, nest)))
        V.append((f'Nesterov t/(t+3) restart, {k}', lambda P, k=k: momentum(P, k, nest, restart=True)))
        V.append((f'Nesterov t/(t+3) restart proj, {k}', lambda P, k=k: momentum(P, k, nest, restart=True, proj=True)))
        for b in (0.3, 0.5, 0.7):
            V.append((f'momentum beta {b}, {k}', lambda P, k=k, b=b: momentum(P, k, lambda t: b)))
            V.append((f'momentum beta {b} restart, {k}', lambda P, k=k, b=b: momentum(P, k, lambda t: b, restart=True)))
        for om in (1.3, 1.6):
            V.append((f'sweep relaxation {om}, {k}', lambda P, k=k, om=om: sweep_relax(P, k, om)))
        V.append((f'Anderson m=1, {k}', lambda P, k=k: anderson(P, k, 1)))
        V.append((f'Anderson m=2, {k}', lambda P, k=k: anderson(P, k, 2)))
        V.append((f'Anderson m=3, {k}', lambda P, k=k: anderson(P, k, 3)))
        V.append((f'NNCG, {k}', lambda P, k=k: nncg(P, k)))
        V.append((f'NNCG + 1 plain, {k}', lambda P, k=k: nncg(P, k, 1)))
    for ko, ki in ((8, 2), (4, 4), (8, 4), (4, 2), (2, 8), (8, 8)):
        V.append((f'cube block outer {ko} x inner {ki}', lambda P, ko=ko, ki=ki: cube_block(P, ko, ki)))
    for name, fn in V:
        report(name, [P.err(fn(P)) for P in Ps], cases)
