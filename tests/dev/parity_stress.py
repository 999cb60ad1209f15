"""One-off stress of the bit-exact parity (GPU box): more envs, more seeds, longer rollouts than the test-suite."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import parity_util as pu
from oracle_util import load_oracle
from leibnizgym_amd import _capi
hip, orc = _capi.load_hip_library(), load_oracle()
n, steps = int(sys.argv[1]), int(sys.argv[2])
for cfg in pu.CONFIGS:
    for seed in (11, 12):
        got = pu.rollout(hip, "cuda:0", n, steps, cfg, seed=seed, episode_length=60)
        want = pu.rollout(orc, "cpu", n, steps, cfg, seed=seed, episode_length=60)
        for t, (a, b) in enumerate(zip(got, want)):
            pu.assert_bit_equal(a, b, f"{cfg} seed {seed} step {t}")
        print(f"{cfg} seed {seed}: {n} envs x {steps} steps bit-identical", flush=True)
