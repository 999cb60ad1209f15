"""GPU suite: the HIP path must reproduce the oracle BIT FOR BIT on every per-env output (state, obs,
states, reward, flags, counters) along seeded rollouts with resets, goal resets and contacts; the eleven
reduced info scalars agree to fp32 summation-order tolerance (rtol 2e-5)."""
import pytest
import torch

import parity_util as pu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("variant", pu.VARIANTS)
@pytest.mark.parametrize("cfg_name", list(pu.CONFIGS))
def test_rollout_bit_exact(hip, oracle, cfg_name, variant):
    n, steps = 1000, 130          # 1000 = 15 full waves + a ragged one; 130 steps > 3 episodes of 40
    got = pu.rollout(hip, DEV, n, steps, cfg_name, variant=variant)
    want = pu.oracle_rollout(oracle, n, steps, cfg_name)
    for t, (a, b) in enumerate(zip(got, want)):
        pu.assert_bit_equal(a, b, f"{cfg_name} [{variant}] step {t}")


def test_both_instantiations_get_the_occupancy_they_are_built_for(hip):
    """The 128-register instantiation is built to put FOUR workgroups on a CU (its LDS is 4 x 39.75 KB of the CU's 160 KB, to the byte), the
    256-register one two: what the HIP runtime computes for the kernels this engine launches - headline, extended DR and box object."""
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    kw = {k: v for k, v in pu.CONFIGS["d4_torque_asym"].items()}
    for extra in ({}, {"domain_randomization": pu.CONFIGS["d4_domain_randomization_extended"]["domain_randomization"]}, {"model": hip.box_model((0.02, 0.08, 0.02), 500.0)}):
        e = TrifingerEngine(make_config(hip, 4096, **{**kw, **extra}), device=DEV, lib=hip)
        e.kernel_variant = "narrow"
        assert e.kernel_occupancy == 4, (extra.keys(), e.kernel_occupancy)
        e.kernel_variant = "wide"
        assert e.kernel_occupancy == 2, (extra.keys(), e.kernel_occupancy)
        e.kernel_variant = "wide_helpers"            # eight wavefronts of 256 registers: one workgroup per CU
        assert e.kernel_variant == "wide_helpers" and e.kernel_occupancy == 1, (extra.keys(), e.kernel_occupancy)
        e.close()


def test_kernel_variant_follows_the_population(hip):
    """tf_create picks the 256-register instantiation up to TF_WIDE_MAX_ENVS envs - with helper wavefronts up to TF_HELPERS_MAX_ENVS when the model
    holds the middle-distal rows (the default) - and the 128-register one above (what bench.py's headline size runs); the override is what the tests
    above use."""
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    for n, want in ((64, "wide_helpers"), (16384, "wide_helpers"), (16385, "wide"), (32768, "wide"), (32769, "narrow"), (65536, "narrow")):
        e = TrifingerEngine(make_config(hip, n, **{k: v for k, v in pu.CONFIGS["d4_torque_asym"].items()}), device=DEV, lib=hip)
        assert e.kernel_variant == want, (n, e.kernel_variant)
        e.kernel_variant = "narrow"
        assert e.kernel_variant == "narrow"
        e.kernel_variant = "auto"
        assert e.kernel_variant == want
        e.close()


@pytest.mark.parametrize("cfg_name,fused_actions,variant", [("d4_torque_asym", False, "narrow"), ("d4_domain_randomization", False, "narrow"), ("d4_torque_asym", True, "narrow"),
                                                            ("d4_torque_asym", False, "wide"), ("d4_domain_randomization", True, "wide"),
                                                            ("d4_torque_asym", True, "wide_helpers"), ("d4_domain_randomization_extended", False, "wide_helpers")])
def test_long_episodes_reach_the_boundary_and_stay_bit_exact(hip, oracle, cfg_name, fused_actions, variant):
    """The 40-step episodes above never let a cube reach the boundary of the arena.  With 750-step episodes under random actions the
    rollout arrives at the steady state of the bench workload - a third of the envs with a live boundary contact - which is where the slot
    order of the boundary corners, the per-slot flags and the branch-free boundary block of the sweeps do their work: HIP and oracle side by
    side, compared every 100 steps on the way there and every 10 steps in the steady state (steps 300-900, across the time-out resets at
    750).  `fused_actions`: the same through tf_step_random - the kernel instantiation bench.py times as `value` (actions drawn inside the
    launch), its action_buf included in the comparison."""
    from leibnizgym_amd import _capi as capi
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    n, steps = 640, 900
    engs = []
    for lib, dev in ((hip, DEV), (oracle, "cpu")):
        kw = dict(pu.CONFIGS[cfg_name])
        kw.pop("_clipping", None)
        engs.append(TrifingerEngine(make_config(lib, n, seed=21, episode_length=750, **kw), device=dev, lib=lib))
    engs[0].kernel_variant = variant
    for e in engs:
        e.reset()
    live, compared = 0.0, 0
    for t in range(steps):
        if fused_actions:
            engs[0].step_random()
            engs[1].step_random()
        else:
            act = pu.actions_for(t, n, engs[0].action_dim, 21)
            engs[0].step(act.to(DEV))
            engs[1].step(act)
        if t % 100 == 99 or (t >= 300 and t % 10 == 9):
            a, b = pu.snapshot(engs[0]), pu.snapshot(engs[1])
            pu.assert_bit_equal(a, b, f"{cfg_name} step {t}")
            compared += 1
            live = max(live, float((b["state"][capi.S_CW_FACE] != 0).mean()))
    for e in engs:
        e.close()
    assert live > 0.1 and compared >= 60, (live, compared)          # the rollout did get to the boundary


@pytest.mark.parametrize("extra", [dict(substeps=1, solver_iterations=4), dict(substeps=3, solver_iterations=1),
                                   dict(dt=0.01, solver_iterations=12, control_decimation=3),
                                   dict(gravity=(0.3, -0.2, -3.7)), dict(normalize_action=False, apply_safety_damping=False),
                                   dict(solver_inner=2), dict(solver_iterations=3, solver_inner=3, substeps=1)])
def test_solver_and_stepping_settings(hip, oracle, extra):
    """Loop bounds and stepping parameters other than the Hydra defaults (the env's own default dict asks for 4 position
    iterations; the reference's tests use control_decimation 5): still bit for bit."""
    want = pu.oracle_rollout(oracle, 300, 50, "envdefault_position", extra=extra)
    for variant in pu.VARIANTS:
        got = pu.rollout(hip, DEV, 300, 50, "envdefault_position", extra=extra, variant=variant)
        for t, (a, b) in enumerate(zip(got, want)):
            pu.assert_bit_equal(a, b, f"{extra} [{variant}] step {t}")


@pytest.mark.parametrize("kind", ["box", "extended_dr"])
def test_middle_distal_pairs_in_the_extended_kernels(hip, oracle, kind):
    """TfModel.ff_middle_pairs through the other two kernel families (the box object: EXT-2; every DR feature: EXT-1), both instantiations"""
    extra = {}
    if kind == "box":
        m = hip.box_model((0.02, 0.08, 0.02), 500.0)
        m.ff_middle_pairs = 1
        extra = {"_model_edit": None, "model": m}
    else:
        extra = {"domain_randomization": pu.CONFIGS["d4_domain_randomization_extended"]["domain_randomization"]}
    want = pu.oracle_rollout(oracle, 300, 50, "ff_middle_pairs", extra=extra)
    for variant in pu.VARIANTS:
        got = pu.rollout(hip, DEV, 300, 50, "ff_middle_pairs", extra=extra, variant=variant)
        for t, (a, b) in enumerate(zip(got, want)):
            pu.assert_bit_equal(a, b, f"ff_middle_pairs + {kind} [{variant}] step {t}")


@pytest.mark.parametrize("difficulty", [-1, 2, 5, 6])
def test_remaining_difficulties_and_reset_modes(hip, oracle, difficulty):
    """Goal samplers of the difficulties the Hydra configs do not use (trifinger_env.py:1211-1243), with success goal
    resets on, the object spawned at its default pose and the robot at a random one."""
    extra = dict(task_difficulty=difficulty, object_reset="default", robot_reset="random",
                 success={"activate": True, "bonus": 10.0, "position_tolerance": 0.08, "orientation_tolerance": 3.2})
    got = pu.rollout(hip, DEV, 200, 40, "envdefault_position", episode_length=15, extra=extra)
    want = pu.rollout(oracle, "cpu", 200, 40, "envdefault_position", episode_length=15, extra=extra)
    for t, (a, b) in enumerate(zip(got, want)):
        pu.assert_bit_equal(a, b, f"difficulty {difficulty} step {t}")
    assert any(s["goal_reset_buf"].any() for s in want), "the scenario should exercise goal resets"


def test_gravity_setter_takes_effect(hip, oracle):
    """tf_set_gravity after creation (IsaacEnvBase's set_sim_params path, env_base.py:175-193) == gravity at creation."""
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    kw = dict(pu.CONFIGS["d4_torque_asym"])
    a = TrifingerEngine(make_config(hip, 128, seed=2, gravity=(0.0, 0.0, -1.62), **kw), device=DEV, lib=hip)
    b = TrifingerEngine(make_config(hip, 128, seed=2, **kw), device=DEV, lib=hip)
    b.set_gravity((0.0, 0.0, -1.62))
    a.reset(), b.reset()
    for t in range(10):
        act = pu.actions_for(t, 128, 9, 2).to(DEV)
        a.step(act), b.step(act)
    assert torch.equal(a.state, b.state) and torch.equal(a.obs, b.obs)
    c = TrifingerEngine(make_config(hip, 128, seed=2, **kw), device=DEV, lib=hip)
    c.reset()
    for t in range(10):
        c.step(pu.actions_for(t, 128, 9, 2).to(DEV))
    assert not torch.equal(a.state, c.state)
    a.close(), b.close(), c.close()


@pytest.mark.parametrize("n", [1, 4, 63, 64, 65])
def test_ragged_sizes(hip, oracle, n):
    want = pu.oracle_rollout(oracle, n, 45, "d4_torque_asym")
    for variant in pu.VARIANTS:
        got = pu.rollout(hip, DEV, n, 45, "d4_torque_asym", variant=variant)
        for t, (a, b) in enumerate(zip(got, want)):
            pu.assert_bit_equal(a, b, f"N={n} [{variant}] step {t}")


@pytest.mark.parametrize("variant", pu.VARIANTS)
@pytest.mark.parametrize("cfg_name", ["envdefault_position", "d4_domain_randomization", "d4_domain_randomization_extended"])
def test_split_path_equals_fused(hip, cfg_name, variant):
    """tf_apply_resets/pre_step/simulate/post_step/finish_step == tf_step on the GPU (also with every
    domain-randomisation feature on: the frame-keyed draws must agree between the two paths, and the hand-over of the DR rows from the cube role to the
    finger roles through LDS must hold in a launch that only simulates - barrier #1b of tf_roles.h)."""
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    n = 777
    kw = dict(pu.CONFIGS[cfg_name])
    engs = [TrifingerEngine(make_config(hip, n, seed=5, episode_length=30, **kw), device=DEV, lib=hip)
            for _ in range(2)]
    for e in engs:
        e.kernel_variant = variant
        e.reset()
    for t in range(70):
        act = pu.actions_for(t, n, 9, 5).to(DEV)
        engs[0].step(act)
        e = engs[1]
        e.action_buf.copy_(act)
        e.apply_resets()
        e.pre_step()
        e.simulate()
        e.post_step()
        e.finish_step()
        torch.cuda.synchronize()
        # rows 66.. (wrench accumulators) are written by the split path only, rows 157.. (samples of the next reset) by the fused step only;
        # info[9] (number of resets) is only counted by the fused kernel
        a, b = pu.snapshot(engs[0]), pu.snapshot(engs[1])
        a["info"][9] = b["info"][9] = 0.0
        pu.assert_bit_equal(a, b, f"split vs fused step {t}", skip_rows=[slice(66, 84), slice(157, 172)])


def test_full_size_properties(hip):
    """BASELINE config 3 size (65536 envs): size-independent invariants over 60 steps."""
    from leibnizgym_amd import _capi as capi
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    n = 65536
    kw = dict(pu.CONFIGS["d4_torque_asym"])
    eng = TrifingerEngine(make_config(hip, n, seed=7, episode_length=25, **kw), device=DEV, lib=hip)
    eng.reset()
    g = torch.Generator(device=DEV).manual_seed(7)
    total_resets = 0.0
    for t in range(60):
        eng.step(torch.rand(n, 9, device=DEV, generator=g) * 2 - 1)
        total_resets += float(eng.info[capi.INFO_NUM_RESETS])
        # the in-kernel fold of the 1024 per-wave partials: the four active reward means add up to the mean reward
        active = [k for k, name in enumerate(capi.REWARD_TERM_ORDER) if pu.D4_REWARDS[name]["activate"]]
        folded = float(sum(eng.info[k] for k in active))
        assert abs(folded - float(eng.reward.double().mean())) < 2e-4 * max(1.0, abs(folded)), (t, folded)
    torch.cuda.synchronize()
    st = eng.state
    assert torch.isfinite(st).all() and torch.isfinite(eng.obs).all() and torch.isfinite(eng.states).all()
    qn = st[capi.S_CUBE_Q:capi.S_CUBE_Q + 4].norm(dim=0)
    assert (qn - 1).abs().max() < 1e-5                                  # unit quaternions
    q = st[capi.S_Q:capi.S_Q + 9]
    lo = torch.tensor([-0.33, 0.0, -2.7] * 3, device=DEV)[:, None]
    hi = torch.tensor([1.0, 1.57, 0.0] * 3, device=DEV)[:, None]
    assert (q >= lo).all() and (q <= hi).all()                          # joint limits
    assert st[capi.S_QD:capi.S_QD + 9].abs().max() <= 10.0 + 1e-4       # velocity limit
    assert st[capi.S_TAU:capi.S_TAU + 9].abs().max() <= 0.36 + 1e-7     # torque limit
    assert st[capi.S_CUBE_P + 2].min() > 0.0325 - 3e-3                  # cube never sinks through the floor
    assert torch.hypot(st[capi.S_CUBE_P], st[capi.S_CUBE_P + 1]).max() < 0.2   # stays inside the arena
    assert int(eng.steps.max()) <= 25 and total_resets == 2 * n         # every env timed out exactly twice
    assert float(eng.info[capi.INFO_NUM_NONFINITE]) == 0.0
    obs_q = eng.obs[:, 0:9].T * (hi - lo) * 0.5 + (hi + lo) * 0.5       # obs is the normalised state
    assert (obs_q - q).abs().max() < 1e-5
    eng.close()


def test_maximum_size_matches_the_oracle_at_the_far_end(hip, oracle):
    """TF_MAX_ENVS (2 Mi envs, 1.25 GB of state rows addressed with 32-bit byte offsets): the last 256 envs of the full
    population equal an oracle shard of exactly those envs (the RNG is keyed by the global env id), step by step; one
    env more is refused."""
    from leibnizgym_amd import _capi as capi
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    n, tail = 2097152, 256
    kw = dict(pu.CONFIGS["d4_domain_randomization"])
    big = TrifingerEngine(make_config(hip, n, seed=9, episode_length=4, **kw), device=DEV, lib=hip)
    ref = TrifingerEngine(make_config(oracle, tail, seed=9, episode_length=4, env_id_offset=n - tail, global_num_envs=n, **kw),
                          device="cpu", lib=oracle)
    big.reset(), ref.reset()
    g = torch.Generator().manual_seed(21)
    for t in range(7):
        act_tail = torch.rand(tail, 9, generator=g) * 2 - 1
        act = torch.zeros(n, 9, device=DEV)
        act[n - tail:] = act_tail.to(DEV)
        big.step(act), ref.step(act_tail)
        for name in ("obs", "states", "reward", "reset_buf", "steps", "reset_count"):
            a = getattr(big, name)[n - tail:].cpu()
            assert torch.equal(a, getattr(ref, name)), (t, name)
        assert torch.equal(big.state[:, n - tail:].cpu(), ref.state), t
        assert float(big.info[capi.INFO_NUM_RESETS]) == (n if t == 4 else 0.0)      # time-out of every env, counted exactly
    assert torch.isfinite(big.reward).all()
    big.close(), ref.close()
    with pytest.raises(ValueError):
        TrifingerEngine(make_config(hip, n + 1, **kw), device=DEV, lib=hip)
