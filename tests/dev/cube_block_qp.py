"""Developer experiment (CPU, fp64), round 4: what would an EXACT solve of the cube block cost?  

tests/dev/pgs_blocks.py shows that the block of all rows that touch the cube (finger-cube + floor + wall), solved exactly once per outer sweep,
is what brings the residual of 8 sweeps under the review's bar.  Here that block is solved by an ACTIVE-SET method instead of inner Gauss-Seidel
sweeps, and the work is counted:

  * with the friction bounds of the tangent rows frozen at the current normal impulses, the block is a strictly convex box-constrained QP in its
    impulses, H = D + G M_c^-1 G^T with D the finger-side compliance of the finger-cube rows (3x3 per contact, zero for floor / wall rows) and G the
    rows' maps of the 6-d cube twist - "block diagonal + rank 6";
  * active set: rows at a bound are fixed, the free rows solve a linear system; because of the structure that system is a 6x6 one in the cube twist
    once the free finger-cube contacts are condensed (K = A^-1) and the free rigid rows are handled as equality constraints (KKT, pseudo-inverse for the
    redundant four-corner floor contact); rows whose solution leaves its box are clamped, clamped rows whose multiplier has the wrong sign are freed;
  * outer loop over the friction bounds (two passes).

Reported: residual after k outer sweeps (as pgs_blocks.py), active-set iterations per block solve, and an operation estimate per block solve
(6x6 Schur complement build + factorisation per iteration, row evaluations) against the ~600 flops of one Gauss-Seidel sweep of the same rows.

   python tests/dev/pgs_cases.py 128 /tmp/pgs_cases.pkl && python tests/dev/cube_block_qp.py /tmp/pgs_cases.pkl"""
import sys, pickle
import numpy as np
from pgs_accel import Prob, plain, report
from pgs_blocks import classify

STAT = {"solves": 0, "iters": [], "flops": []}


def block_solve_active_set(P, lam, v, blk, max_iter=40):
    """exact solve of the rows `blk` (all touch the cube; finger-cube rows also their finger) given the other rows' impulses.  Dense fp64 linear algebra on
    the free set - the point is the iteration count and the sizes, not the implementation."""
    n = len(blk)
    J = P.J[blk]                                   # n x 15
    W = J @ P.Minv @ J.T                           # Delassus block (finger + cube parts)
    lam_b = lam[blk].copy()
    r0 = J @ v - W @ lam_b                         # relative velocity of the block rows with the block's own impulses removed
    bias = np.where(P.kind[blk] == 0, P.bias[blk], 0.0)
    kind = P.kind[blk]
    parent = np.array([blk.index(P.parent[i]) if P.kind[i] == 1 else -1 for i in blk])
    mu = P.mu[blk]
    iters_total, flops = 0, 0.0
    x = lam_b.copy()
    for bound_pass in range(3):                    # friction bounds from the current normals, then re-solve
        lo = np.where(kind == 0, 0.0, 0.0); hi = np.where(kind == 0, np.inf, 0.0)
        for i in range(n):
            if kind[i] == 1:
                lim = mu[i] * max(x[parent[i]], 0.0); lo[i], hi[i] = -lim, lim
        x = np.clip(x, lo, hi)
        free = (x > lo) & (x < hi) | ((kind == 0) & (x > 0))
        for it in range(max_iter):
            iters_total += 1
            F = np.where(free)[0]; B = np.where(~free)[0]
            if len(F):
                rhs = -(r0[F] + bias[F] + W[np.ix_(F, B)] @ x[B])
                WFF = W[np.ix_(F, F)]
                xF = np.linalg.lstsq(WFF, rhs, rcond=1e-12)[0]       # redundant corner contacts: minimum-norm solution
                # work estimate with the structure: condense the free finger-cube rows (3x3 inverse each), 6x6 Schur complement from the free rows
                # (21 entries x |F| multiply-adds), Cholesky (6^3 / 3), back-substitution for |F| rows (12 each)
                flops += 2 * (21 * len(F) + 72 + 12 * len(F)) + 30 * (len(F) // 3)
            else:
                xF = np.zeros(0)
            xn = x.copy(); xn[F] = xF
            viol = (xn < lo - 1e-12) | (xn > hi + 1e-12)
            if viol.any():
                # step to the first blocking bound, clamp it (primal active-set step)
                d = xn - x
                alpha, blk_i = 1.0, -1
                for i in F:
                    if d[i] < 0 and xn[i] < lo[i]:
                        a = (lo[i] - x[i]) / d[i]
                    elif d[i] > 0 and xn[i] > hi[i]:
                        a = (hi[i] - x[i]) / d[i]
                    else:
                        continue
                    if a < alpha: alpha, blk_i = a, i
                x = x + alpha * d
                if blk_i >= 0:
                    x[blk_i] = lo[blk_i] if d[blk_i] < 0 else hi[blk_i]; free[blk_i] = False
                continue
            x = xn
            # multipliers of the rows at a bound: relative velocity with the right sign?
            r = r0 + bias + W @ x
            flops += 2 * 6 * n                          # row evaluations through the twist
            rel = np.zeros(n, bool)
            for i in B:
                if x[i] <= lo[i] + 1e-15 and r[i] < -1e-12 and hi[i] > lo[i]: rel[i] = True      # at the lower bound but the row wants more impulse
                if x[i] >= hi[i] - 1e-15 and r[i] > 1e-12 and hi[i] > lo[i] and kind[i] == 1: rel[i] = True
            if not rel.any(): break
            worst = np.argmax(np.where(rel, np.abs(r), 0.0)); free[worst] = True
    STAT["solves"] += 1; STAT["iters"].append(iters_total); STAT["flops"].append(flops)
    dl = x - lam_b
    lam[blk] = x
    v += dl @ (J @ P.Minv)
    return lam, v


def outer_exact(P, k):
    fc, fing, cube = classify(P)
    blk = fc + cube
    lam = np.zeros(P.n); v = P.v0.copy()
    for _ in range(k):
        if blk: lam, v = block_solve_active_set(P, lam, v, blk)
        P.sweep(lam, v, fing)
    return v


if __name__ == '__main__':
    cases = pickle.load(open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/pgs_cases.pkl', 'rb'))
    Ps = [Prob(c) for c in cases]
    for k in (8, 16): report(f'plain PGS (the spec), {k}', [P.err(plain(P, k)) for P in Ps], cases)
    for k in (2, 4, 8):
        STAT["solves"] = 0; STAT["iters"] = []; STAT["flops"] = []
        report(f'cube block by active set, outer {k}', [P.err(outer_exact(P, k)) for P in Ps], cases)
        it = np.array(STAT["iters"]); fl = np.array(STAT["flops"])
        print(f"      block solves {STAT['solves']}: active-set iterations per solve median {np.median(it):.0f} p90 {np.percentile(it, 90):.0f} max {it.max()}; "
              f"operation estimate per solve median {np.median(fl):.0f} p90 {np.percentile(fl, 90):.0f} max {fl.max():.0f} flops (one Gauss-Seidel sweep of the block: ~600)")
