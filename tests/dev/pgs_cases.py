"""Developer experiment (CPU): dump the contact problems of random persistent-contact states as data (rows, M^-1, start velocity, fp64
fixed point) so that solver variants can be compared quickly: tests/dev/pgs_accel.py.   python tests/dev/pgs_cases.py [cases] [out.pkl]"""
import sys, os, pickle
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'tests')); sys.path.insert(0, REPO)
import numpy as np, torch
from multiprocessing import Pool


def one(seed):
    import test_contact_lcp_reference as L
    import physics_ref as PR, test_physics_analytic as T
    from oracle_util import load_oracle
    torch.set_num_threads(1)
    lib = load_oracle(); rng = np.random.default_rng(seed)
    H = L.H
    for _ in range(50):
        q, qd, cube, tau = L.make_case(rng)
        eng = T.engine(lib, device='cpu', dt=H, substeps=1, solver_iterations=8)
        f32 = dict(dtype=torch.float32)
        eng.q[:, 0] = torch.tensor(q, **f32); eng.qd[:, 0] = torch.tensor(qd, **f32); eng.cube[:, 0] = torch.tensor(cube, **f32); eng.tau[:, 0] = torch.tensor(tau, **f32)
        for _ in range(4): eng.simulate()
        st = eng.state[:, 0].numpy().astype(np.float64); eng.close()
        try: ref = PR.ref_substep(st[0:9], st[9:18], st[18:31], tau, H, max_sweeps=50000)
        except ValueError: continue
        det = ref[3]
        fc = [x for x in det['fc'] if x[3].lam > 0]; te = [x for x in det['te'] if x[3].lam > 0]
        if det['sweeps'] >= 50000 or not (fc or te): continue
        R = det['rows']
        rows = dict(J=np.array([r.J for r in R]), kind=np.array([{'normal': 0, 'tangent': 1, 'limit': 2}[r.kind] for r in R]),
                    bias=np.array([r.bias for r in R]), mu=np.array([r.mu for r in R]), lo=np.array([r.lo for r in R]), hi=np.array([r.hi for r in R]),
                    parent=np.array([R.index(r.parent) if r.parent is not None else -1 for r in R]), lam=np.array([r.lam for r in R]))
        vstar = np.concatenate(ref[:3])
        return dict(rows=rows, Minv=det['Minv'], v0=det['v_start'], vstar=vstar, nfloor=det['n_floor'], nwall=det['n_wall'], nfc=len(fc), nte=len(te), sweeps=det['sweeps'], seed=seed)
    return None


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    out = sys.argv[2] if len(sys.argv) > 2 else '/tmp/pgs_cases.pkl'
    with Pool(8) as p: R = [r for r in p.map(one, range(2000, 2000 + n)) if r]
    pickle.dump(R, open(out, 'wb'))
    print(len(R), 'cases ->', out)
