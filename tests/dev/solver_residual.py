"""Developer checker (CPU, oracle; under tests/ because only tests/ may use oracle/): TRUNCATION residual of the shipped sweep count in the product's own arithmetic - the substep with
`sweeps` solver iterations against the same substep (same state, same warm-start rows) iterated 4096 times, scaled as in
tests/test_contact_lcp_reference.py (joints / 10 rad/s, cube m/s, cube rad/s / 20).  Thousands of states in seconds (no fp64 reference
in the loop: what is measured is the solver, not the modelling differences between the spec and its fp64 restatement).
State sets:  cases  - the random persistent-contact states of tests/test_contact_lcp_reference.make_case after `k_warm` product substeps
             rollout - states of the bench workload (random actions, 65536-env statistics on a sample) every 50 steps
   python tests/dev/solver_residual.py cases|rollout [n] [sweeps ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import numpy as np, torch


def gen_cases(n, seed=20261003):
    import test_contact_lcp_reference as L
    path = f"/tmp/solver_residual_cases_{n}_{seed}.npz"
    if os.path.exists(path):
        d = np.load(path); return d["q"], d["qd"], d["cube"], d["tau"]
    from multiprocessing import Pool
    with Pool(8) as p: out = p.map(_gen_chunk, [(seed + 1000 * k, n // 8 + 1) for k in range(8)])
    q, qd, cube, tau = (np.concatenate([o[i] for o in out])[:n] for i in range(4))
    np.savez(path, q=q, qd=qd, cube=cube, tau=tau)
    return q, qd, cube, tau


def _gen_chunk(args):
    import test_contact_lcp_reference as L
    seed, n = args
    rng = np.random.default_rng(seed)
    cs = [L.make_case(rng) for _ in range(n)]
    return tuple(np.array([c[i] for c in cs]) for i in range(4))


def scaled(a, b):
    """per-env scaled velocity difference of two state tensors [rows, n]"""
    d = (a - b).abs()
    return torch.maximum(torch.maximum(d[9:18].max(0).values / 10.0, d[25:28].max(0).values), d[28:31].max(0).values / 20.0).numpy()


def live_finger_contact(lam):
    """[n] bool: a finger-cube or fingertip-floor contact carries an impulse (rows from TF_S_LAM_FC on: 12 finger-cube rows, 3 activity
    codes - the finger-cube rows of a finger with code & 3 == 0 are undefined -, 9 fingertip-floor rows)"""
    code = lam[12:15].to(torch.int32) & 3
    fc = ((code != 0) & (lam[0:12:4] > 0)).any(0)
    return fc | (lam[15:24:3] > 0).any(0)


def stats(name, e):
    print(f"{name:58s} n {len(e):6d}  median {np.median(e):.2e}  p90 {np.percentile(e, 90):.2e}  p99 {np.percentile(e, 99):.2e}  p99.9 {np.percentile(e, 99.9):.2e}  max {e.max():.2e}", flush=True)


def main():
    import test_physics_analytic as T
    import test_contact_lcp_reference as L
    from oracle_util import load_oracle
    from leibnizgym_amd import _capi as capi
    lib = load_oracle()
    try:
        lib.dll.tfo_omp_threads(8)
    except Exception:
        pass
    mode = sys.argv[1] if len(sys.argv) > 1 else "cases"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    sweeps_list = [int(s) for s in sys.argv[3:]] or [8]
    REF = 4096
    if mode == "cases":
        q, qd, cube, tau = gen_cases(n)
        f32 = dict(dtype=torch.float32)
        for sw in sweeps_list:
            for k_warm in (4, 20):
                eng = T.engine(lib, n=n, dt=L.H, substeps=1, solver_iterations=sw)
                eng.q.copy_(torch.tensor(q.T, **f32)); eng.qd.copy_(torch.tensor(qd.T, **f32)); eng.cube.copy_(torch.tensor(cube.T, **f32)); eng.tau.copy_(torch.tensor(tau.T, **f32))
                for _ in range(k_warm): eng.simulate()
                saved = eng.state.clone()
                eng.simulate(); got = eng.state.clone()
                ref = T.engine(lib, n=n, dt=L.H, substeps=1, solver_iterations=REF)
                ref.state.copy_(saved); ref.simulate(); want = ref.state.clone()
                eng.state.copy_(saved); eng.state[capi.S_LAM_FC:] = 0.0; eng.simulate(); cold = eng.state.clone()
                ref.state.copy_(saved); ref.state[capi.S_LAM_FC:] = 0.0; ref.simulate(); want_c = ref.state.clone()
                # contact present: some warm-start impulse is non-zero in the converged solution
                lam = want[capi.S_LAM_FC:]
                has = live_finger_contact(lam)
                stats(f"cases k_warm {k_warm:2d}, {sw:2d} sweeps, warm (finger contact live)", scaled(got, want)[has.numpy()])
                stats(f"cases k_warm {k_warm:2d}, {sw:2d} sweeps, cold (finger contact live)", scaled(cold, want_c)[has.numpy()])
                eng.close(); ref.close()
    else:
        import bench
        kw = bench.workload_kwargs(True)
        from leibnizgym_amd.engine import TrifingerEngine, make_config
        for sw in sweeps_list:
            kw["solver_iterations"] = sw
            eng = TrifingerEngine(make_config(lib, n, seed=7, **kw), device="cpu", lib=lib)
            kr = dict(kw); kr["solver_iterations"] = REF; kr["substeps"] = 1; kr["dt"] = 0.01
            ref = TrifingerEngine(make_config(lib, n, seed=7, **kr), device="cpu", lib=lib)
            k1 = dict(kw); k1["substeps"] = 1; k1["dt"] = 0.01
            one = TrifingerEngine(make_config(lib, n, seed=7, **k1), device="cpu", lib=lib)
            g = torch.Generator().manual_seed(3)
            eng.reset()
            errs, hasc = [], []
            for step in range(1, 601):
                eng.step(torch.rand(n, 9, generator=g) * 2 - 1)
                if step % 50 == 0:
                    saved = eng.state.clone()
                    one.state.copy_(saved); one.simulate(); got = one.state.clone()
                    ref.state.copy_(saved); ref.simulate(); want = ref.state.clone()
                    lam = want[capi.S_LAM_FC:]
                    has = live_finger_contact(lam)
                    errs.append(scaled(got, want)); hasc.append(has.numpy())
            e = np.concatenate(errs); h = np.concatenate(hasc)
            stats(f"rollout (random actions), {sw:2d} sweeps, all envs", e)
            stats(f"rollout (random actions), {sw:2d} sweeps, envs with a live finger contact ({100.0 * h.mean():.0f} %)", e[h])
            eng.close(); ref.close(); one.close()


if __name__ == "__main__":
    main()
